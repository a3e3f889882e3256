"""Interleaved A/B of ENGINE arrangements (module globals of painter_amd.engine) against the whole training step (ViT-L, B = 8, bf16,
train mode), one process, one box.

    python tools/step_engine_ab.py ROUNDS STEPS  "name:GLOBAL=value[,GLOBAL=value...]"  ["name2:..." ...]

e.g.  python tools/step_engine_ab.py 4 6 "delta in dQ:_ATTN_PREP=fused" "prep launch:_ATTN_PREP=launch"
Globals not named by a setting keep their start-up value.  Round 5 uses it for: Delta inside the dQ kernel vs the prep launch, fc1's bias
gradient from the GEMM epilogue vs a separate column-sum pass."""
import statistics
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from painter_amd import engine, models_painter  # noqa: E402


def main():
    rounds, steps = int(sys.argv[1]), int(sys.argv[2])
    settings = []
    for spec in sys.argv[3:]:
        name, kv = spec.split(":")
        settings.append((name, dict(a.split("=") for a in kv.split(",") if a)))
    dev = torch.device("cuda")
    m = models_painter.painter_vit_large_patch16_input896x448(compute_dtype="bf16")
    bench.randomize_parameters(m, seed=1)
    m = m.to(dev).train()
    c = m._cfg
    inp = bench.synthetic_inputs(8, c.H, c.W, c.L, 1234, dev)

    repack = [False]           # pseudo-global REPACK=1: drop the packed rel-pos tables before every step (= a step after an optimizer update)

    def step():
        for p in m.parameters():
            p.grad = None
        if repack[0]:
            m._hot.relpos_stale()
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
    touched = sorted({k for _, kn in settings for k in kn if k != "REPACK"})
    saved = {k: getattr(engine, k) for k in touched}
    res = {name: [] for name, _ in settings}
    try:
        for _ in range(rounds):
            for name, kn in settings:
                for k in touched:
                    setattr(engine, k, kn.get(k, saved[k]))
                repack[0] = kn.get("REPACK", "0") == "1"
                m._hot._rcache.clear()                  # (the two pack arrangements keep different things in it)
                res[name].append(timed())
    finally:
        for k, v in saved.items():
            setattr(engine, k, v)
    for name, ts in res.items():
        print("%-44s median %.3f ms  min %.3f  (%s)" % (name, statistics.median(ts), min(ts), " ".join("%.2f" % t for t in ts)), flush=True)


if __name__ == "__main__":
    main()
