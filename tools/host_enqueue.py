"""How long does the host need to ENQUEUE one training step (forward + backward, ViT-L, B = 8) compared with what the GPU needs to run
it?  If the two are close, the step is at the mercy of host jitter; a hipGraph of the step would be the remedy.  Diagnostics."""
import sys
import time

import torch

sys.path.insert(0, ".")
import bench                                                    # noqa: E402
from painter_amd import models_painter                          # noqa: E402


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
        t0 = time.perf_counter()
        loss, _, _ = model(imgs, tgts, bool_masked_pos=mask, valid=valid)
        t1 = time.perf_counter()
        loss.backward()
        t2 = time.perf_counter()
        return t1 - t0, t2 - t1

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    res = []
    for _ in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        f, b = step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        res.append((f * 1e3, b * 1e3, (t1 - t0) * 1e3, (t2 - t0) * 1e3))
    for r in res:
        print("host enqueue: forward %.1f ms, backward %.1f ms, total %.1f ms   |   step complete after %.1f ms" % r, flush=True)


if __name__ == "__main__":
    main()
