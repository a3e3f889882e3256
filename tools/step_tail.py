"""How long does the side stream (weight gradients) run after the data-gradient chain has finished?  Records an event on each stream at
the end of the backward and prints backward start -> main done / side done, for several weight-gradient workgroup targets.  Diagnostics."""
import sys

import torch

sys.path.insert(0, ".")
import bench                                                    # noqa: E402
from painter_amd import models_painter                          # noqa: E402
from painter_amd._lib import lib                                # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    model = models_painter.painter_vit_large_patch16_input896x448(compute_dtype="bf16")
    bench.randomize_parameters(model, seed=1)
    model = model.to(dev).train()
    cfg = model._cfg
    imgs, tgts, mask, valid = bench.synthetic_inputs(8, cfg.H, cfg.W, cfg.L, 1234, dev)
    hp = model._hot
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    hp.tail_probe = (ev[2], ev[3])
    for target in (128, 96, 160, 192, 256, 128):
        lib.pa_debug_set(3, target)
        res = []
        for it in range(6):
            for p in model.parameters():
                p.grad = None
            ev[0].record()
            loss, _, _ = model(imgs, tgts, bool_masked_pos=mask, valid=valid)
            ev[1].record()
            loss.backward()
            ev[4].record()
            torch.cuda.synchronize()
            if it >= 2:
                res.append((ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2]), ev[1].elapsed_time(ev[3]), ev[0].elapsed_time(ev[4])))
        m = [sum(r[i] for r in res) / len(res) for i in range(4)]
        print("wgrad workgroup target %3d: forward %.2f ms | backward: main chain done %.2f, side stream done %.2f | step %.2f ms"
              % (target, m[0], m[1], m[2], m[3]), flush=True)


if __name__ == "__main__":
    main()
