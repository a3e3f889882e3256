"""Interleaved A/B of two BUILDS of the library against the whole training step (ViT-L, B = 8, bf16, train mode), one box.

    python tools/step_lib_ab.py ROUNDS STEPS  "name=path/to/libA.so"  "name2=path/to/libB.so" ...

Each round starts one process per build (PAINTER_AMD_LIB selects the library; tools/step_knob_ab.py does the timing: 3 x STEPS steps
between HIP events after warm-up) in alternation, so that clock drift of the box lands on both builds alike.  Prints median and minimum
per build.  Round 5 uses it for source-level changes that have no run-time knob (the epilogue column layout of the fp32 outputs)."""
import os
import re
import statistics
import subprocess
import sys


def main():
    rounds, steps = int(sys.argv[1]), sys.argv[2]
    builds = [s.split("=", 1) for s in sys.argv[3:]]
    res = {n: [] for n, _ in builds}
    for _ in range(rounds):
        for name, path in builds:
            env = dict(os.environ, PAINTER_AMD_LIB=os.path.abspath(path))
            out = subprocess.run([sys.executable, "tools/step_knob_ab.py", "3", steps, "x:"], env=env, capture_output=True, text=True, timeout=400)
            m = re.search(r"\(([\d. ]+)\)", out.stdout)
            if not m:
                print(name, "FAILED", out.stdout[-300:], out.stderr[-600:], flush=True)
                continue
            res[name] += [float(v) for v in m.group(1).split()]
    for name, ts in res.items():
        if ts:
            print("%-40s median %.3f ms  min %.3f  (%s)" % (name, statistics.median(ts), min(ts), " ".join("%.2f" % t for t in ts)), flush=True)


if __name__ == "__main__":
    main()
