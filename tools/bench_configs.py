"""Training throughput of the other BASELINE configs on ONE GPU (cfg 3: ResNeXt-50-FPN 800^2 bs 2,
cfg 4: DenseNet-121-FPN 640^2 bs 4), eager + hipGraph.  Not the headline bench (that is bench.py)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "retinanet-tensorflow_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
import bench


def run(backbone, size, batch, steps=8, warmup=3, use_graph=True):
    import dataset, layers, levels, retinanet, train
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    torch.manual_seed(0)
    lv = levels.build_levels()
    net = retinanet.RetinaNet(backbone, lv, 80, layers.elu, 0.2).to(dev)
    rng = np.random.default_rng(0)
    image = torch.randn(batch, size, size, 3, device=dev)
    boxes = np.zeros((batch, 32, 4), np.float32); cls = np.zeros((batch, 32), np.int32); nobj = np.zeros(batch, np.int32)
    for i in range(batch):
        b, c, o = bench.synthetic_objects(rng, size)
        boxes[i], cls[i], nobj[i] = b, c, o
    boxes, cls, nobj = (torch.from_numpy(a).to(dev) for a in (boxes, cls, nobj))

    def features():
        c, r, m = dataset.build_labels((size, size), cls, boxes, lv, 80, num_obj=nobj)
        return {"image": image, "detection": {"classifications": c, "regressions": r}, "trainable_masks": m}
    features.concurrent = False        # (the assignment of `batch` independent images on the main stream, as before)

    trainer = train.Trainer(net, lv, loss_mode="focal", device=dev, use_graph=use_graph, input_fn=features)
    for _ in range(warmup):
        trainer.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        trainer.step()
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    trainer.check_device_errors()
    # the gradient all-reduce schedule a multi-GPU run would follow: heads + FPN after segment A, then one slice per backbone
    # part (last stage first); only the LAST part's slice is reduced after the last backward kernel
    ranges = list(trainer._graphs[2]) if (use_graph and trainer._graphs) else []
    total = 4 * trainer.arena.count
    after = 4 * (ranges[-1][1] - ranges[-1][0]) if ranges else 4 * trainer.cut_offset
    return {"backbone": backbone, "image_size": size, "batch": batch, "images_per_sec": round(batch * steps / el, 2),
            "ms_per_step": round(1e3 * el / steps, 2), "hip_graph": use_graph,
            "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
            "backward_parts": 1 + len(ranges), "gradient_bytes": total, "bytes_after_backward": after,
            "bytes_after_backward_frac": round(after / total, 4),
            "allreduce_slices_MB": [round(4 * (trainer.arena.count - trainer.cut_offset) / 1e6, 1)] + [round(4 * (hi - lo) / 1e6, 1) for lo, hi in ranges]}


if __name__ == "__main__":
    cfgs = (("resnet_50", 800, 2), ("densenet_121", 640, 4), ("mobilenet_v2", 512, 2))
    only = sys.argv[1] if len(sys.argv) > 1 else None        # e.g. `densenet_121`: that config alone (for a rocprofv3 run)
    for cfg in cfgs:
        if only is None or cfg[0] == only:
            print(json.dumps(run(*cfg)), flush=True)
