"""Interleaved sweep of the process-wide tuning knobs against the whole training step (ViT-L, B = 8, bf16, train mode):
    pa_debug_set(3, n)   workgroup target of the weight-gradient GEMMs that run on the side stream beside the data-gradient chain
    pa_debug_set(6, s)   K splits of the rel-pos table-gradient GEMM there
Rounds of [setting: n steps timed], median per setting.  python tools/knob_sweep.py [rounds] [steps]"""
import statistics
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from painter_amd import models_painter  # noqa: E402
from painter_amd._lib import lib  # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    dev = torch.device("cuda")
    m = models_painter.painter_vit_large_patch16_input896x448(compute_dtype="bf16")
    bench.randomize_parameters(m, seed=1)
    m = m.to(dev).train()
    c = m._cfg
    inp = bench.synthetic_inputs(8, c.H, c.W, c.L, 1234, dev)

    def step():
        for p in m.parameters():
            p.grad = None
        loss, _, _ = m(inp[0], inp[1], bool_masked_pos=inp[2], valid=inp[3])
        loss.backward()

    def timed():
        step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            step()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / steps

    for _ in range(3):
        step()
    import os
    wgs = [int(v) for v in os.environ.get("SWEEP_WGS", "96,128,160,192,256").split(",")]
    settings = [("wgrad WGs %d" % n, 3, n) for n in wgs] + ([("rel-pos splits %d" % s, 6, s) for s in (2, 4, 8)] if "SWEEP_WGS" not in os.environ else [])
    res = {k: [] for k, _, _ in settings}
    for _ in range(rounds):
        for name, which, val in settings:
            lib.pa_debug_set(3, 128)
            lib.pa_debug_set(6, 4)
            lib.pa_debug_set(which, val)
            res[name].append(timed())
    lib.pa_debug_set(3, 128)
    lib.pa_debug_set(6, 4)
    for name, _, _ in settings:
        v = statistics.median(res[name])
        print("%-22s %.2f ms/step = %.1f images/s   %s" % (name, v, 8e3 / v, ["%.2f" % t for t in res[name]]), flush=True)


if __name__ == "__main__":
    main()
