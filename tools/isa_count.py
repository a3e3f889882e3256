"""Instruction mix per basic block of the kernels in a gfx950 assembly file (hipcc --save-temps ... -> *-gfx950.s).

    python tools/isa_count.py file.s [kernel-name-substring]

Prints, for every block that contains matrix instructions, the count of MFMA / VALU (with exp, cvt, permlane/DPP split out) / LDS /
global / scalar / waitcnt+barrier instructions, and the issue-cycle estimate (MFMA 32x32x16 = 8 passes of 4 cycles; VALU 4 cycles
per wave64 instruction, transcendental 8).  Used to find what the inner loops of the attention kernels spend their issue slots on."""
import re
import sys
from collections import Counter, OrderedDict


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt")):
        return "trans"
    if op.startswith("v_cvt"):
        return "cvt"
    if op.startswith(("v_permlane", "v_readlane", "v_readfirstlane", "v_writelane")) or "dpp" in op:
        return "xlane"
    if op.startswith("v_accvgpr"):
        return "acc_mov"
    if op.startswith("v_pk_"):
        return "vpk"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith(("s_waitcnt", "s_barrier", "s_nop", "s_sleep", "s_setprio", "s_sched")):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    kernel, block = None, None
    data = OrderedDict()
    for line in open(path):
        line = line.rstrip()
        m = re.match(r"^([A-Za-z_][\w$.]*):", line)
        if m and not line.startswith(".L"):
            kernel, block = m.group(1), "entry"
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", line)
        if m:
            block = m.group(1)
            continue
        if kernel is None or want not in kernel:
            continue
        t = line.strip()
        if not t or t.startswith((";", ".", "//")):
            continue
        op = t.split()[0]
        if op == "s_endpgm":
            data.setdefault((kernel, block), Counter())
            kernel = None
            continue
        c = data.setdefault((kernel, block), Counter())
        c[classify(op)] += 1
        if "dpp" in t and classify(op) != "xlane":
            c["xlane"] += 1
            c[classify(op)] -= 1
        if op.startswith("s_cbranch") or op == "s_branch":
            c["->" + t.split()[-1]] += 1
    last = None
    for (k, b), c in data.items():
        if c["mfma"] == 0 and sum(v for kk, v in c.items() if not kk.startswith("->")) < 40:
            continue
        if k != last:
            print("\n== " + k)
            last = k
        valu = c["valu"] + c["cvt"] + c["xlane"] + c["vpk"] + c["acc_mov"]
        cyc_v = 4 * valu + 8 * c["trans"]
        br = " ".join(kk for kk in c if kk.startswith("->"))
        print("%-12s mfma %3d (%5d cyc) | valu %4d cvt %3d xlane %3d acc_mov %3d pk %3d trans %3d (%5d cyc) | lds %3d vmem %3d salu %3d wait %3d  %s"
              % (b, c["mfma"], 32 * c["mfma"], c["valu"], c["cvt"], c["xlane"], c["acc_mov"], c["vpk"], c["trans"], cyc_v, c["lds"], c["vmem"], c["salu"], c["wait"], br))


if __name__ == "__main__":
    main()
