"""Round-4 interleaved A/B of the three removals against the whole training step (ViT-L, B = 8, bf16, train mode), one process, one box:
    pa_debug_set(4, 1 | 0)   256-row GEMM tiles everywhere | the 224-row tile where it fills the last round better
    pa_debug_set(7, 1 | 2)   rel-pos table gradient through dG + gather GEMM | contracted inside the dQ kernel
    pa_debug_set(8, 1 | 2)   attention workgroup order: light workgroups interleaved | dispatched last
Rounds of [setting: n steps timed], median per setting; "all off" / "all on" bracket the list.  python tools/r04_ab.py [rounds] [steps]"""
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
    OFF, ON = {4: 1, 7: 1, 8: 1}, {4: 0, 7: 2, 8: 2}
    def with_(base, which, val):
        d = dict(base)
        d[which] = val
        return d
    settings = [("all off (round-3 arrangement)", OFF), ("224-row GEMM tile by rule only", with_(OFF, 4, 0)), ("224-row tile everywhere", with_(OFF, 4, 2)),
                ("fused rel-pos gradient only", with_(OFF, 7, 2)), ("light workgroups last only", with_(OFF, 8, 2)), ("all on", ON)]
    res = {k: [] for k, _ in settings}
    for _ in range(rounds):
        for name, knobs in settings:
            for which, val in knobs.items():
                lib.pa_debug_set(which, val)
            res[name].append(timed())
    for which in (4, 7, 8):
        lib.pa_debug_set(which, 0)
    for name, _ in settings:
        v = statistics.median(res[name])
        print("%-32s %.2f ms/step = %.1f images/s   %s" % (name, v, 8e3 / v, ["%.2f" % t for t in res[name]]), flush=True)


if __name__ == "__main__":
    main()
