"""Interleaved A/B of the whole training step (ViT-L, B = 8, bf16, train mode) between attention generations in ONE process:
rounds of [generation x: n steps timed] for x in the list.  python tools/step_ab.py [rounds] [steps] [gens e.g. 2,3,0]"""
import sys
import time

import torch

sys.path.insert(0, ".")
import bench                                                    # noqa: E402
from painter_amd import models_painter                           # noqa: E402
from painter_amd._lib import lib                                 # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    gens = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [2, 3, 0]
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
        return loss

    for g in gens:
        lib.pa_attn_set_generation(g)
        step()
    res = {g: [] for g in gens}
    for _ in range(rounds):
        for g in gens:
            lib.pa_attn_set_generation(g)
            step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize()
            res[g].append((time.perf_counter() - t0) / steps * 1e3)
    lib.pa_attn_set_generation(0)
    for g in gens:
        print("generation %d: ms/step %s  min %.2f  (%.1f images/s)" % (g, ["%.2f" % t for t in res[g]], min(res[g]), 8e3 / min(res[g])), flush=True)


if __name__ == "__main__":
    main()
