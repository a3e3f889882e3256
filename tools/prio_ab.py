"""Do HIP stream priorities help the two-stream backward?  The data-gradient chain (main stream) and the parameter-gradient kernels (side
stream) share the CUs; with equal priority the dispatcher interleaves their workgroups.  Arrangements, interleaved in one process
(ViT-L, B = 8, bf16, train mode):  main default / side default (the product arrangement);  main HIGH priority / side default;
main default / side LOW (if the runtime has a level below the default).  python tools/prio_ab.py [rounds] [steps]"""
import statistics
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from painter_amd import models_painter  # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    dev = torch.device("cuda")
    m = models_painter.painter_vit_large_patch16_input896x448(compute_dtype="bf16")
    bench.randomize_parameters(m, seed=1)
    m = m.to(dev).train()
    c = m._cfg
    inp = bench.synthetic_inputs(8, c.H, c.W, c.L, 1234, dev)
    print("stream priorities torch reports for new streams: default %d, priority=-1 -> %d, priority=1 -> %s"
          % (torch.cuda.Stream().priority, torch.cuda.Stream(priority=-1).priority, _try(lambda: torch.cuda.Stream(priority=1).priority)), flush=True)

    def step():
        for p in m.parameters():
            p.grad = None
        loss, _, _ = m(inp[0], inp[1], bool_masked_pos=inp[2], valid=inp[3])
        loss.backward()

    def timed(main_stream):
        with torch.cuda.stream(main_stream):
            step()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(steps):
                step()
            e1.record()
            torch.cuda.synchronize()
        return e0.elapsed_time(e1) / steps

    cur = torch.cuda.current_stream()
    hi = torch.cuda.Stream(priority=-1)
    cases = [("main default, side default", cur, 0), ("main HIGH, side default", hi, 0), ("main default, side priority 1 (low)", cur, 1),
             ("main HIGH, side HIGH", hi, -1)]
    res = {k: [] for k, _, _ in cases}
    for _ in range(3):
        step()
    for _ in range(rounds):
        for name, ms, sp in cases:
            m._hot.side_priority = sp
            m._hot._side = {}
            try:
                res[name].append(timed(ms))
            except Exception as e:      # a priority the runtime refuses
                res[name].append(float("nan"))
                print(name, "failed:", e)
    for name, _, _ in cases:
        v = statistics.median(res[name])
        print("%-40s %.2f ms/step = %.1f images/s   %s" % (name, v, 8e3 / v, ["%.2f" % t for t in res[name]]), flush=True)


def _try(fn):
    try:
        return fn()
    except Exception as e:
        return "refused (%s)" % type(e).__name__


if __name__ == "__main__":
    main()
