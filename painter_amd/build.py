"""Build libpainter_hip.so (gfx950) in-tree with plain hipcc: one object per .hip, linked into
painter_amd/lib/libpainter_hip.so.  No torch headers, no hipify, no JIT cache -- the .so travels with the
source snapshot to the GPU box.  `python -m painter_amd.build [--force]`."""
import concurrent.futures as cf
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "lib", "obj" + ("" if "PA_LIB_NAME" not in os.environ else "_" + os.environ["PA_LIB_NAME"].replace(".", "_")))
LIB = os.path.join(HERE, "lib", os.environ.get("PA_LIB_NAME", "libpainter_hip.so"))    # PA_LIB_NAME / PA_EXTRA_FLAGS: A/B experiment builds
ARCH = "gfx950"
# -fno-slp-vectorize: hipcc (ROCm 7.2) SLP-packs adjacent fp32 adds / muls / fmas into v_pk_*_f32.  On gfx950 a kernel built that
# way (LayerNorm backward: the packed (s1, s2) row-sum accumulators) returned a wrong s2 for one row in ~1 % of its launches whenever
# MFMA workgroups of ANOTHER kernel were resident on the same CU (second HIP stream; tools/race_iso.py reproduces it in isolation,
# 32 / 3600 launches, 0 / 3600 without the flag), which made the two-stream backward non-deterministic.  Without SLP packing the
# whole step is bit-stable in every probe and not slower (125.7 vs 126.4 images/s).  PA_SLP=1 re-enables it for experiments.
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Wno-unused-result"]
if os.environ.get("PA_SLP") != "1":
    FLAGS = FLAGS + ["-fno-slp-vectorize"]
FLAGS = FLAGS + os.environ.get("PA_EXTRA_FLAGS", "").split()
# seggpt_io.hip / pair_io.hip reproduce host float arithmetic (numpy, Pillow) bit for bit: no fused multiply-add there.
EXTRA = {"seggpt_io.hip": ["-ffp-contract=off"], "pair_io.hip": ["-ffp-contract=off"]}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _digest(paths):
    h = hashlib.sha256()
    for p in sorted(paths):
        with open(p, "rb") as f:
            h.update(os.path.basename(p).encode())
            h.update(f.read())
    return h.hexdigest()


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(os.path.dirname(HERE), "include", "painter_hip.h"))
    return sorted(hs)


def _compile(src, stamp):
    obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
    tag = obj + ".stamp"
    extra = EXTRA.get(os.path.basename(src), [])
    want = _digest([src] + headers()) + "|" + " ".join(extra) + "|" + " ".join(FLAGS)
    if not stamp and os.path.exists(obj) and os.path.exists(tag) and open(tag).read() == want:
        return obj, False
    cmd = [_hipcc()] + FLAGS + extra + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr[-4000:]))
    with open(tag, "w") as f:
        f.write(want)
    return obj, True


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sources()
    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        res = list(ex.map(lambda s: _compile(s, force), srcs))
    objs = [o for o, _ in res]
    # prune objects / stamps whose source is gone (a retired kernel file would otherwise ride along in the shipped snapshot)
    keep = {os.path.basename(o) for o in objs}
    for f in os.listdir(OBJ):
        if f.endswith(".o") and f not in keep or f.endswith(".o.stamp") and f[:-6] not in keep:
            os.remove(os.path.join(OBJ, f))
    if force or any(c for _, c in res) or not os.path.exists(LIB):
        cmd = [_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr[-4000:])
        if verbose:
            print("built", LIB)
    elif verbose:
        print("up to date", LIB)
    _record_build_info()
    return LIB


def _record_build_info():
    """painter_amd/lib/build_info.json: the git head this tree was built at and a digest of the kernel sources.  The snapshot that goes to
    the GPU box has no .git; bench.py reads this file there to identify the build in its JSON line (refreshed on every build() call)."""
    import json
    root = os.path.dirname(HERE)
    head = None
    try:
        r = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True, timeout=10)
        if r.returncode == 0 and r.stdout.strip():
            d = subprocess.run(["git", "-C", root, "status", "--porcelain", "--untracked-files=no"], capture_output=True, text=True, timeout=10)
            head = r.stdout.strip() + ("-dirty" if d.stdout.strip() else "")
    except Exception:
        pass
    if head is None:
        return                                   # no git here (the GPU box): keep what the build container recorded
    with open(os.path.join(HERE, "lib", "build_info.json"), "w") as f:
        json.dump({"git_head": head, "source_digest": _digest(sources() + headers())[:16]}, f)


if __name__ == "__main__":
    build(force="--force" in sys.argv)
