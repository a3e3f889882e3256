"""Build-container half of tests/test_reference_engine_gpu.py: the reference's unmodified drivers import through the stubs of
oracle/ref_import.py, and everything they touch on the model exists on the painter_amd classes (SURVEY.md 8b 'Attributes/methods
callers touch').  Skipped where no reference checkout is mounted."""
import ast
import os

import pytest

from oracle import ref_import

pytestmark = pytest.mark.skipif(not ref_import.reference_available(), reason="/root/reference not mounted")


def _model_attrs(path, names=("model", "model_without_ddp")):
    """Attribute names read off `model` / `model.module` in a reference driver (static scan)."""
    tree = ast.parse(open(path).read())
    found = set()
    for node in ast.walk(tree):
        if isinstance(node, ast.Attribute):
            v = node.value
            if isinstance(v, ast.Name) and v.id in names:
                found.add(node.attr)
            if isinstance(v, ast.Attribute) and v.attr == "module" and isinstance(v.value, ast.Name) and v.value.id in names:
                found.add(node.attr)
    return found


def test_reference_drivers_import_and_touch_only_attributes_our_modules_have():
    eng = ref_import.load_reference_engine_train()
    assert callable(eng.train_one_epoch) and callable(eng.evaluate_pt)
    assert hasattr(eng.misc, "NativeScalerWithGradNormCount") and hasattr(eng.misc, "MetricLogger") and hasattr(eng.lr_sched, "adjust_learning_rate")
    seg = ref_import.load_reference_seggpt_engine()
    assert callable(seg.run_one_image) and callable(seg.inference_image)
    from painter_amd import models_painter, models_seggpt
    deepspeed_only = {"backward", "step", "optimizer"}          # engine_train.py:75-76: the `loss_scaler is None` (DeepSpeed) branch
    used = _model_attrs(os.path.join(ref_import.PAINTER_DIR, "engine_train.py")) - {"module"} - deepspeed_only
    used_seg = _model_attrs(os.path.join(ref_import.SEGGPT_DIR, "seggpt_engine.py")) - {"module"}
    import torch.nn as nn
    probe = models_painter.Painter.__dict__.keys() | nn.Module.__dict__.keys() | {"patch_size", "patch_embed"}
    missing = sorted(a for a in used if a not in probe)
    assert not missing, missing
    probe_seg = probe | models_seggpt.SegGPT.__dict__.keys() | {"seg_type"}
    missing = sorted(a for a in used_seg if a not in probe_seg)
    assert not missing, missing
    assert {"unpatchify", "patch_size"} <= used | used_seg          # the scan does see what the drivers use


def test_run_one_image_expectation_of_the_gpu_test_holds_for_the_reference_model_itself():
    """tests/test_reference_engine_gpu.py compares the unmodified `run_one_image` on OUR SegGPT module with a picture computed from the CPU
    oracle.  Here the same comparison is made with the REFERENCE's own model in the driver (CPU, tiny config): it pins the test's
    expectation, so a failure on a GPU box can only come from our module."""
    import torch

    from oracle import painter_oracle as O
    from tests.golden.make_golden import build_reference
    eng = ref_import.load_reference_seggpt_engine()
    cfg = O.tiny_config(seggpt=True)
    model, P = build_reference(cfg, 5)
    model.eval()
    model.seg_type = "instance"
    imgs, tgts, _, _ = O.synthetic_batch(cfg, 2, 9, "half")
    out = eng.run_one_image(imgs.permute(0, 2, 3, 1).double().numpy(), tgts.permute(0, 2, 3, 1).double().numpy(), model, torch.device("cpu"))
    L = cfg.grid[0] * cfg.grid[1]
    mask = torch.zeros(1, L)
    mask[:, L // 2:] = 1
    with torch.no_grad():
        _, yo, _ = O.forward(P, cfg, imgs, tgts, mask, torch.ones_like(tgts), torch.ones(2, 1), 0)
    y = O.unpatchify(yo, cfg.patch_size).permute(0, 2, 3, 1)
    ref = torch.clip((y[0, y.shape[1] // 2:] * torch.tensor(O.IMAGENET_STD) + torch.tensor(O.IMAGENET_MEAN)) * 255, 0, 255)
    assert out.shape == ref.shape and float((out - ref).abs().max()) < 1e-2


@pytest.mark.skipif(not os.path.isdir("/root/reference/Painter"), reason="needs the mounted reference tree to compare against")
def test_staged_reference_subset_is_byte_identical_to_the_reference_and_is_what_the_gpu_box_imports(tmp_path, monkeypatch):
    """oracle/stage_ref.py (run by __graft_entry__.build()): the archive under the git-ignored oracle/_ref/ holds byte-for-byte copies of
    exactly the listed reference files, unpacks into a directory the loader accepts as a reference root, and a process that cannot see
    /root/reference (the GPU box) imports the drivers from there."""
    import hashlib
    import subprocess
    import sys

    from oracle import stage_ref
    assert stage_ref.stage(verbose=False) == len(stage_ref.FILES)
    root = stage_ref.unpack()
    for rel in stage_ref.FILES:
        a = open(os.path.join("/root/reference", rel), "rb").read()
        b = open(os.path.join(root, rel), "rb").read()
        assert hashlib.sha256(a).hexdigest() == hashlib.sha256(b).hexdigest(), rel
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # nothing of the reference is tracked by git: the staging area is ignored as a whole
    out = subprocess.run(["git", "-C", repo, "check-ignore", "oracle/_ref/reference_subset.tar.gz"], capture_output=True, text=True)
    assert out.returncode == 0
    code = ("import os, sys; sys.path.insert(0, %r)\n"
            "real = os.path.isfile\n"
            "os.path.isfile = lambda p: False if str(p).startswith('/root/reference') else real(p)\n"
            "from oracle import ref_import as r\n"
            "assert r.reference_available() and not r.REFERENCE_ROOT.startswith('/root/reference'), r.REFERENCE_ROOT\n"
            "e = r.load_reference_engine_train(); s = r.load_reference_seggpt_engine(); m = r.load_reference_painter()\n"
            "assert callable(e.train_one_epoch) and callable(s.run_one_image) and hasattr(m, 'Painter')\n"
            "print('ok', r.REFERENCE_ROOT)\n") % repo
    env = dict(os.environ)
    env.pop("PAINTER_REFERENCE_ROOT", None)
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert res.returncode == 0 and res.stdout.startswith("ok"), res.stderr[-2000:]
