"""How fast does the shapes stream train?  The product's loop (DeviceFeed + hipGraph step) at 256^2, mAP on 32 held-out images at
several step counts -- the source of the numbers pinned in tests/test_gpu_train_cli.py::test_shapes_training_reaches_a_pinned_map."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "retinanet-tensorflow_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch


def main(loss="bce_dice", marks=(300, 600, 1000, 1500, 2500, 4000), lr=1e-2):
    import train
    from data_loaders.shapes import Shapes
    from test_gpu_train_cli import _shapes_trainer
    dev = torch.device("cuda:0")
    net, tr, feed, lv = _shapes_trainer(dev, True, True, dropout=0.2, seed=0, scale=256, loss=loss)
    tr.opt.lr = lr
    done, t0 = 0, time.perf_counter()
    for m in marks:
        while done < m:
            out = tr.step(); done += 1
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        res = train.evaluate(net, Shapes(None, image_size=(320, 256), seed=12345), lv, 32, scale=256, device=dev)
        print(json.dumps({"loss": loss, "lr": lr, "steps": done, "train_s": round(el, 2), "class_loss": round(out["class_loss"].item(), 4),
                          "regr_loss": round(out["regr_loss"].item(), 4), "mAP": round(res["mAP"], 4), "AP50": round(res["AP50"], 4),
                          "AP75": round(res["AP75"], 4), "class_iou": round(res["class_iou"], 4), "regr_iou": round(res["regr_iou"], 4)}), flush=True)
        t0 = time.perf_counter() - el
    feed.close()


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "bce_dice", lr=float(sys.argv[2]) if len(sys.argv) > 2 else 1e-2,
         marks=(300, 600, 1000, 1500, 2500) if len(sys.argv) > 2 else (300, 600, 1000, 1500, 2500, 4000))
