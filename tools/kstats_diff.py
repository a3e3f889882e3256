"""Side-by-side of two rocprofv3 --stats kernel summaries (…_kernel_stats.csv), per step: python tools/kstats_diff.py new.csv old.csv [steps_new steps_old]"""
import csv
import sys


def load(p):
    rows = {}
    lines = [ln for ln in open(p) if not ln.startswith("#")]
    for r in csv.DictReader(lines):
        rows[r["Name"]] = (int(r["Calls"]), float(r["TotalDurationNs"]), float(r["AverageNs"]))
    return rows


new, old = load(sys.argv[1]), load(sys.argv[2])
sn = float(sys.argv[3]) if len(sys.argv) > 3 else 8.0
so = float(sys.argv[4]) if len(sys.argv) > 4 else 6.0
print("kernel time per step: new %.2f ms  old %.2f ms" % (sum(v[1] for v in new.values()) / 1e6 / sn, sum(v[1] for v in old.values()) / 1e6 / so))
for n in sorted(set(new) | set(old), key=lambda n: -(new.get(n, (0, 0, 0))[1] / sn + old.get(n, (0, 0, 0))[1] / so))[:int(sys.argv[5]) if len(sys.argv) > 5 else 45]:
    a, b = new.get(n), old.get(n)
    f = lambda v, s: "%7.1f us x %5.1f = %6.2f ms" % (v[2] / 1e3, v[0] / s, v[1] / 1e6 / s) if v else " " * 34
    print("%s | %s | %s" % (f(a, sn), f(b, so), n[:120]))
