"""Turns the two rocprofv3 PMC passes of tools/attn_bench.py (gpu_visit.sh pmcattn: matrix-pipe / VALU / wait counters, then the LDS
counters) into per-kernel utilisation fractions.  Normalisation on MI355X: GRBM_GUI_ACTIVE is summed over the 8 XCDs (kernel cycles =
value / 8), SQ_VALU_MFMA_BUSY_CYCLES over 1024 SIMDs, SQ_LDS_IDX_ACTIVE / SQ_LDS_BANK_CONFLICT over 256 CUs, SQ_ACTIVE_INST_VALU counts
quad-cycles per SIMD, SQ_WAIT_* / SQ_WAVE_CYCLES are per-wave quad-cycle sums (used as ratios only).
    python tools/pmc_attn_summary.py gpurun_out/pmcattn_SQ_VALU_MFMA_BUSY_CYCLES gpurun_out/pmcattn_SQ_LDS_IDX_ACTIVE > profiles/r03_attn_sq_counters.json"""
import collections
import csv
import json
import os
import sys


def load(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    rows = list(csv.DictReader(open(os.path.join(d, "pmc_counter_collection.csv"))))
    first = rows[0]["Counter_Name"]
    for r in rows:
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == first:
            n[k] += 1
    return {k: {c: v / n[k] for c, v in cs.items()} for k, cs in agg.items()}, n


def main():
    a, na = load(sys.argv[1])
    b, _ = load(sys.argv[2])
    out = {}
    for k in a:
        if not (k.startswith("a3::") or k.startswith("a2::")) or "prep" in k or "etab" in k:
            continue
        cyc = a[k]["GRBM_GUI_ACTIVE"] / 8.0
        r = {"launches": na[k], "kernel_cycles": round(cyc),
             "mfma_busy": round(a[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / cyc, 3),
             "valu_busy": round(a[k]["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / cyc, 3),
             "wave_wait_any": round(a[k]["SQ_WAIT_ANY"] / a[k]["SQ_WAVE_CYCLES"], 3),
             "wave_issue_stall": round(a[k]["SQ_WAIT_INST_ANY"] / a[k]["SQ_WAVE_CYCLES"], 3)}
        if k in b:
            r["lds_array_active"] = round(b[k]["SQ_LDS_IDX_ACTIVE"] / 256.0 / cyc, 3)
            r["lds_bank_conflict_share"] = round(b[k]["SQ_LDS_BANK_CONFLICT"] / max(b[k]["SQ_LDS_IDX_ACTIVE"], 1.0), 3)
            r["wave_lds_issue_stall"] = round(b[k]["SQ_WAIT_INST_LDS"] / b[k]["SQ_WAVE_CYCLES"], 3)
        out[k] = r
    json.dump({"workload": "tools/attn_bench.py (ViT-L shape, B' = 8 and 16 mixed; generation 3 = a3::, generation 2 = a2::)",
               "git_head": os.environ.get("PAINTER_AMD_GIT_HEAD"), "kernels": out}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
