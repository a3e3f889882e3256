"""Per-kernel matrix-pipe / VALU utilisation from one rocprofv3 PMC pass of bench.py (SQ_VALU_MFMA_BUSY_CYCLES, SQ_ACTIVE_INST_VALU,
SQ_BUSY_CYCLES, SQ_WAVE_CYCLES, SQ_WAIT_ANY, SQ_WAIT_INST_ANY, GRBM_GUI_ACTIVE; separate run, --kernel-trace only):
    python tools/pmc_mfma.py <dir with *counter_collection.csv> <out.csv>
MFMA busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles); kernel cycles = GRBM_GUI_ACTIVE / 8 (the counter is summed over
the 8 XCDs).  SQ_ACTIVE_INST_VALU counts quad-cycles (x 4).  SURVEY.md 8(d) "Evidence"."""
import collections
import csv
import glob
import sys


def main():
    src, out = sys.argv[1], sys.argv[2]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    for f in glob.glob(src + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                calls[k] += 1
    rows = []
    for k, c in agg.items():
        cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        if cyc <= 0:
            continue
        rows.append((cyc, k, calls[k], c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024 * cyc), 4 * c.get("SQ_ACTIVE_INST_VALU", 0.0) / (1024 * cyc),
                     c.get("SQ_WAIT_ANY", 0.0) / max(c.get("SQ_WAVE_CYCLES", 1.0), 1.0), c.get("SQ_WAIT_INST_ANY", 0.0) / max(c.get("SQ_WAVE_CYCLES", 1.0), 1.0)))
    rows.sort(reverse=True)
    with open(out, "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "launches", "kernel_cycles_total", "mfma_busy_frac", "valu_busy_frac", "wave_wait_any_frac", "wave_wait_inst_frac"])
        for cyc, k, n, mf, vf, wa, wi in rows:
            w.writerow([k[:140], n, int(cyc), "%.4f" % mf, "%.4f" % vf, "%.4f" % wa, "%.4f" % wi])
    for cyc, k, n, mf, vf, wa, wi in rows[:16]:
        print("%-90s n=%4d mfma %.3f valu %.3f wait %.3f/%.3f" % (k[:90], n, mf, vf, wa, wi))


if __name__ == "__main__":
    main()
