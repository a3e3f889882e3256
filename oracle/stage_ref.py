"""TEST INFRASTRUCTURE ONLY -- stages the handful of UNMODIFIED reference files the parity tests and bench.py's `cpu_baseline` need
into oracle/_ref/ so that they travel to the GPU box (which has no /root/reference).

    python -m oracle.stage_ref            (also called by __graft_entry__.build() whenever /root/reference is present)

What is staged (byte-for-byte, relative paths kept, as ONE archive oracle/_ref/reference_subset.tar.gz with a MANIFEST.json of the
members' SHA-256 beside it; oracle/ref_import.py unpacks it on first use into a directory the process creates for itself -- mkdtemp, 0700, removed at exit):
  Painter/models_painter.py, Painter/engine_train.py, Painter/util/{misc,lr_sched,vitdet_utils,masking_generator,lr_decay}.py,
  SegGPT/SegGPT_inference/{models_seggpt.py, seggpt_engine.py, util/vitdet_utils.py}
i.e. the model classes (the oracle behind `cpu_baseline.kind == "reference"`: Painter.forward, models_painter.py:464-472) and the two
drivers tests/test_reference_engine_gpu.py runs on the HIP modules (engine_train.train_one_epoch :34-144, seggpt_engine.run_one_image
:26-53).  oracle/_ref/ is listed in .gitignore -- the copies never enter this repository's history -- but not in .gpurunignore, so the
snapshot that goes to the GPU box carries them, like the built .so files.  oracle/ref_import.py falls back to oracle/_ref/ when
neither PAINTER_REFERENCE_ROOT nor /root/reference exists.  Nothing under painter_amd/ imports any of this."""
import hashlib
import io
import json
import os
import sys
import tarfile
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
DEST = os.path.join(HERE, "_ref")
FILES = [
    "Painter/models_painter.py",
    "Painter/engine_train.py",
    "Painter/util/misc.py",
    "Painter/util/lr_sched.py",
    "Painter/util/lr_decay.py",
    "Painter/util/vitdet_utils.py",
    "Painter/util/masking_generator.py",
    "SegGPT/SegGPT_inference/models_seggpt.py",
    "SegGPT/SegGPT_inference/seggpt_engine.py",
    "SegGPT/SegGPT_inference/util/vitdet_utils.py",
]


ARCHIVE = os.path.join(DEST, "reference_subset.tar.gz")
MANIFEST = os.path.join(DEST, "MANIFEST.json")


def stage(src_root="/root/reference", verbose=True):
    """-> number of files staged (0 when src_root has no reference checkout: nothing is touched then)."""
    if not os.path.isfile(os.path.join(src_root, FILES[0])):
        if verbose:
            print("oracle.stage_ref: no reference checkout at %s -- keeping whatever oracle/_ref/ holds" % src_root)
        return 0
    os.makedirs(DEST, exist_ok=True)
    manifest = {}
    with tarfile.open(ARCHIVE, "w:gz") as tar:
        for rel in FILES:
            data = open(os.path.join(src_root, rel), "rb").read()
            manifest[rel] = hashlib.sha256(data).hexdigest()
            info = tarfile.TarInfo(rel)
            info.size = len(data)
            info.mtime = 0                       # reproducible archive
            tar.addfile(info, io.BytesIO(data))
    with open(MANIFEST, "w") as f:
        json.dump({"source": src_root, "sha256": manifest}, f, indent=1, sort_keys=True)
    if verbose:
        print("oracle.stage_ref: %d unmodified reference files -> %s" % (len(FILES), ARCHIVE))
    return len(FILES)


_UNPACKED = None


def unpack():
    """-> directory holding the staged files, or None when nothing is staged.  Unpacked ONCE PER PROCESS into a directory this process
    created itself (tempfile.mkdtemp: mode 0700, unpredictable name, removed at exit) and checked against MANIFEST.json after writing --
    ref_import puts the directory on sys.path, so it must not be a predictable, pre-creatable path in the shared temp dir."""
    global _UNPACKED
    if _UNPACKED is not None:
        return _UNPACKED
    if not (os.path.isfile(ARCHIVE) and os.path.isfile(MANIFEST)):
        return None
    import atexit
    import shutil
    man = json.load(open(MANIFEST))["sha256"]
    root = tempfile.mkdtemp(prefix="painter_amd_reference_subset_")
    atexit.register(shutil.rmtree, root, ignore_errors=True)
    with tarfile.open(ARCHIVE, "r:gz") as tar:
        for m in tar.getmembers():
            if m.name not in man or not m.isfile():
                raise RuntimeError("oracle/_ref: unexpected archive member %r" % m.name)
            data = tar.extractfile(m).read()
            if hashlib.sha256(data).hexdigest() != man[m.name]:
                raise RuntimeError("oracle/_ref: %s does not match MANIFEST.json" % m.name)
            dst = os.path.join(root, m.name)
            os.makedirs(os.path.dirname(dst), exist_ok=True)
            with open(dst, "wb") as f:
                f.write(data)                       # the very bytes that were hashed
    if sorted(man) != sorted(rel for rel in man if os.path.isfile(os.path.join(root, rel))):
        raise RuntimeError("oracle/_ref: archive is missing files of MANIFEST.json")
    _UNPACKED = root
    return root


if __name__ == "__main__":
    n = stage(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
    sys.exit(0 if n or unpack() else 1)
