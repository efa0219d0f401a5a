"""BASELINE configs[4] on one GPU: ResNeXt-50-FPN 1024x1024, batch 16, forward (training=False) + sigmoid (inside the candidate scan) +
anchor decode + batched class-wise NMS, in fp32 and in fp16 storage (f16 matrix-core convs, fp32 accumulate)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "retinanet-tensorflow_amd")):
    sys.path.insert(0, p)
import torch


def main(backbone="resnet_50", size=1024, batch=16, iters=5):
    import layers, levels, ops, retinanet, utils
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    lv = levels.build_levels()
    net = retinanet.RetinaNet(backbone, lv, 80, layers.elu, 0.0).to(dev)
    image = torch.randn(batch, size, size, 3, device=dev)
    anchors = {k: lv[k].normalized_anchor_sizes((size, size)) for k in lv}

    def run():
        with torch.no_grad():
            out = net(image, training=False)
            rows = sum(v.numel() // 80 for v in out["classifications"].values())
            # the class logits and box deltas go to the detector as the net wrote them (fp16 in fp16 mode); the sigmoid runs
            # inside the candidate scan
            return utils.detect_raw(out["classifications"], out["regressions"], anchors, 80, score_threshold=0.0105,
                                    capacity=int(rows * 0.5), return_raw=True, logits=True)

    for dtype in ("f32", "f16"):
        layers.set_inference_dtype(dtype)
        o = run(); torch.cuda.synchronize()
        counts = o[5].cpu().tolist()
        t0 = time.perf_counter()
        for _ in range(iters):
            run()
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / iters
        # the same pass replayed from ONE hipGraph (static input buffer, static outputs: what a serving loop at a fixed shape does)
        graph_ips = None
        if os.environ.get("RN_INF_GRAPH", "1") == "1":
            try:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    og = run()
                g.replay(); torch.cuda.synchronize()
                assert og[5].cpu().tolist()[:2] == counts[:2], "graph replay: candidate / survivor counts differ from the eager pass"
                t0 = time.perf_counter()
                for _ in range(iters):
                    g.replay()
                torch.cuda.synchronize()
                graph_ips = round(batch * iters / (time.perf_counter() - t0), 2)
                del g, og
            except Exception as e:       # (reported, not fatal: the eager figure stands)
                graph_ips = "failed: %s" % (str(e)[:120],)
        print(json.dumps({"backbone": backbone, "image_size": size, "batch": batch, "dtype": dtype,
                          "images_per_sec": round(batch / el, 2), "ms_per_batch": round(el * 1e3, 2),
                          "images_per_sec_graph": graph_ips,
                          "candidates": counts[0], "kept": counts[1], "conv_TFLOPs": round(596.0 * batch / el / 1e3, 1),
                          "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}), flush=True)
    layers.set_inference_dtype("f32")


if __name__ == "__main__":
    main()
