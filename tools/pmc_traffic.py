"""Turns the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only) of `bench.py` into
profiles/roofline_traffic.json: HBM bytes per launch of the roofline kernels.
Units / corrections (MI355X_MICROARCH.md, section HBM): FETCH_SIZE and WRITE_SIZE are reported in KiB-like units of 1024 B
by rocprofv3's derived metric (TCC_EA0_RDREQ x 64 B / 1024); on gfx950 FETCH_SIZE counts wide (16 B/lane) streaming reads at
half their size, so the read figure is doubled.  WRITE_SIZE is used as reported (uncalibrated, see the guide)."""
import collections
import csv
import glob
import json
import sys

# every substring of a tuple has to occur in the kernel name (the mangled dgrad name carries the ILV template argument in the middle)
KERNELS = {"fc1": ("Epi4BiasGelu",), "attn_fwd": ("a3::fwd_kernel",), "wgrad": ("Epi4Slab",), "dgrad": ("gemm256_kernelILb0ELb1E", "8Epi4BiasIDF16b"),
           "attn_bwd_dq": ("a3::bwd_dq_kernel",), "attn_bwd_dkv": ("a3::bwd_dkv_kernel",)}


def per_launch(path, counter):
    vals = collections.defaultdict(list)
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            for key, pat in KERNELS.items():
                if all(p_ in r["Kernel_Name"] for p_ in pat):
                    vals[key].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
    return vals


def main():
    fetch_dir, write_dir, out = sys.argv[1], sys.argv[2], sys.argv[3]
    fe, wr = per_launch(fetch_dir, "FETCH_SIZE"), per_launch(write_dir, "WRITE_SIZE")
    res = {}
    for key in KERNELS:
        if not fe.get(key) or not wr.get(key):
            continue
        # the most common grid size = the B=8 launches of blocks 3..23 (grid sizes of the 2B-wide blocks differ)
        grid = collections.Counter(g for g, _ in fe[key]).most_common(1)[0][0]
        f = [v for g, v in fe[key] if g == grid]
        w = [v for g, v in wr[key] if g == grid]
        fk, wk = sum(f) / len(f), sum(w) / len(w)
        res[key] = {"grid_size_threads": grid, "launches_sampled": len(f), "FETCH_SIZE_raw_KB": round(fk, 1), "WRITE_SIZE_raw_KB": round(wk, 1),
                    "hbm_read_bytes": int(2 * fk * 1024), "hbm_write_bytes": int(wk * 1024),
                    "hbm_bytes_per_launch": int(2 * fk * 1024 + wk * 1024),
                    "note": "read = 2 x FETCH_SIZE (gfx950 wide-read correction), write = WRITE_SIZE as reported"}
    # which build the counters belong to: bench.py refuses the file when the library it loads has another hash
    import hashlib
    import os
    sys.path.insert(0, ".")
    from painter_amd._lib import LIB_PATH
    res["_meta"] = {"lib_sha16": hashlib.sha256(open(LIB_PATH, "rb").read()).hexdigest()[:16],
                    "git_head": os.environ.get("PAINTER_AMD_GIT_HEAD") or (json.load(open("painter_amd/lib/build_info.json")).get("git_head") if os.path.exists("painter_amd/lib/build_info.json") else None),
                    "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python bench.py --steps 2 --warmup 1"}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
