"""Interleaved A/B of library knobs (pa_debug_set) against the whole training step (ViT-L, B = 8, bf16, train mode), one process, one box.

    python tools/step_knob_ab.py ROUNDS STEPS  "name:knob=value[,knob=value...]"  ["name2:..." ...]

e.g.  python tools/step_knob_ab.py 4 6 "light last on:8=2" "light last off:8=1"
Rounds of [setting: STEPS steps timed with HIP events], median and min per setting; every knob touched is restored afterwards
(pa_debug_get).  Round 5 uses it for the `light attention workgroups last` decision (VERDICT round 4, item 2 ii)."""
import statistics
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from painter_amd import models_painter  # noqa: E402
from painter_amd._lib import lib  # noqa: E402


def main():
    rounds, steps = int(sys.argv[1]), int(sys.argv[2])
    settings = []
    for spec in sys.argv[3:]:
        name, kv = spec.split(":")
        settings.append((name, {int(a.split("=")[0]): int(a.split("=")[1]) for a in kv.split(",") if a}))
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
    touched = sorted({k for _, kn in settings for k in kn})
    saved = {k: lib.pa_debug_get(k) for k in touched}
    res = {name: [] for name, _ in settings}
    try:
        for _ in range(rounds):
            for name, knobs in settings:
                for k in touched:
                    lib.pa_debug_set(k, knobs.get(k, saved[k]))
                res[name].append(timed())
    finally:
        for k, v in saved.items():
            lib.pa_debug_set(k, v)
    for name, ts in res.items():
        print("%-40s median %.3f ms  min %.3f  (%s)" % (name, statistics.median(ts), min(ts), " ".join("%.2f" % t for t in ts)), flush=True)


if __name__ == "__main__":
    main()
