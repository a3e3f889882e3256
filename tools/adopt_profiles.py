"""Files measurements of a GPU visit (gpurun_out/...) under profiles/ -- stamped with the library they were taken on, and ONLY when that
library is the one this tree builds right now.

Rounds 4 and 5 closed with kernel statistics taken one commit before the shipped library ("Missing 5"): the artefacts
trailed the code.  `tools/gpu_visit.sh` now records the hash of the library it ran (gpurun_out/visit_lib_sha16.txt); this script compares
it with the hash of painter_amd/lib/libpainter_hip.so as built here and refuses on a mismatch, so a profile can only enter profiles/ while
the tree still builds the binary it describes.  tests/test_bench_cpu.py checks the stamps of the closing artefacts against one another and
against the built library.

    python tools/adopt_profiles.py "<what was run>" gpurun_out/prof_one_stream/one_kernel_stats.csv=profiles/r06_bench_one_stream_kernel_stats.csv ...
    python tools/adopt_profiles.py --allow-stale "<what was run>" src=dst ...     (mid-round evidence of an EARLIER library: stamped with the visit's hash)
"""
import hashlib
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def lib_sha16():
    with open(os.path.join(ROOT, "painter_amd", "lib", "libpainter_hip.so"), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def main():
    args = sys.argv[1:]
    allow_stale = "--allow-stale" in args
    args = [a for a in args if a != "--allow-stale"]
    what, pairs = args[0], [a.split("=", 1) for a in args[1:]]
    visit = open(os.path.join(ROOT, "gpurun_out", "visit_lib_sha16.txt")).read().split()
    vsha, vhead = visit[0], (visit[1] if len(visit) > 1 else "unknown")
    here = lib_sha16()
    if vsha != here and not allow_stale:
        sys.exit("REFUSED: the visit measured library %s, this tree builds %s -- re-run the visit on the current build (or --allow-stale for "
                 "mid-round evidence, which is then stamped with the visit's hash)" % (vsha, here))
    for src, dst in pairs:
        src, dst = os.path.join(ROOT, src), os.path.join(ROOT, dst)
        if dst.endswith(".csv") or dst.endswith(".log") or dst.endswith(".txt"):
            with open(src) as f:
                body = f.read()
            with open(dst, "w") as f:
                f.write("# library %s git %s | %s\n" % (vsha, vhead, what))
                f.write(body)
        else:
            shutil.copyfile(src, dst)       # JSON artefacts carry their own stamp (bench.py: build.lib_sha16; pmc_traffic.py: _meta.lib_sha16)
        print("filed %s (library %s)" % (os.path.relpath(dst, ROOT), vsha))


if __name__ == "__main__":
    main()
