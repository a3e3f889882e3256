"""Which stream is the backward's critical path?  Times the training step (ViT-L, B = 8, bf16) while moving pieces of the side stream's
work to the main stream.  Diagnostics."""
import os
import sys
import time

import torch

sys.path.insert(0, ".")
import bench                                                    # noqa: E402
from painter_amd import engine, models_painter                  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    model = models_painter.painter_vit_large_patch16_input896x448(compute_dtype="bf16")
    bench.randomize_parameters(model, seed=1)
    model = model.to(dev).train()
    cfg = model._cfg
    imgs, tgts, mask, valid = bench.synthetic_inputs(8, cfg.H, cfg.W, cfg.L, 1234, dev)

    def step():
        for p in model.parameters():
            p.grad = None
        loss, _, _ = model(imgs, tgts, bool_masked_pos=mask, valid=valid)
        loss.backward()

    def timed(n=5):
        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    allw = {"dec", "fc2", "fc1", "proj", "qkv"}
    cases = [("default (all weight gradients + column sums + rel-pos on the side stream)", None, True),
             ("column sums on the main stream", allw | {"nocolsum"}, True),
             ("rel-pos / conv weight gradients on the main stream", None, False),
             ("column sums and rel-pos on the main stream", allw | {"nocolsum"}, False),
             ("proj weight gradient on the main stream too", {"dec", "fc2", "fc1", "qkv"}, True),
             ("one stream", "off", True)]
    for _ in range(2):
        for name, filt, extra in cases:
            hp = model._hot
            hp.use_side_stream = filt != "off"
            hp.side_filter = None if filt in (None, "off") else filt
            engine._SIDE_EXTRA = extra
            print("%-85s %.2f ms/step" % (name, timed()), flush=True)


if __name__ == "__main__":
    main()
