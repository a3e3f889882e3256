"""TEST INFRASTRUCTURE ONLY -- loader for the *unmodified* reference (build container only).

Imports /root/reference/Painter/models_painter.py and
/root/reference/SegGPT/SegGPT_inference/models_seggpt.py under private module names after
injecting minimal stand-ins for the third-party packages that are absent from this image
(timm==0.3.2, detectron2, fvcore, fairscale).  Used by
tests/golden/make_golden.py to generate the committed golden vectors and by the CPU tests that
pin oracle/painter_oracle.py against the real reference when /root/reference is mounted.

/root/reference does NOT exist on the GPU box.  There the loader falls back to oracle/_ref/reference_subset.tar.gz: byte-for-byte
copies of the few reference files the drivers need, staged by oracle/stage_ref.py at build time (git-ignored, so they never enter this
repository; the gpurun snapshot carries the archive like the built .so files) and unpacked into a temporary directory on first use.  Product code (painter_amd/) never imports anything from oracle/.

Stub semantics (timm 0.3.2, pinned at Painter/requirements.txt:1, asserted main_train.py:24):
  * Mlp       = fc1 -> act_layer() -> Dropout(drop) -> fc2 -> Dropout(drop)
  * DropPath  = x.div(keep) * floor(keep + rand([B,1,..,1]))   in training, identity otherwise
  * trunc_normal_ == torch.nn.init.trunc_normal_
The detectron2 / fvcore / fairscale symbols are only referenced by the never-instantiated
ResBottleneckBlock and the disabled activation-checkpoint hook (models_painter.py:92-150,
:311-312); trivial stand-ins are enough.
"""
import importlib.util
import os
import sys
import types

import torch
import torch.nn as nn

def _reference_root():
    env = os.environ.get("PAINTER_REFERENCE_ROOT")
    if env:
        return env
    if os.path.isfile("/root/reference/Painter/models_painter.py"):
        return "/root/reference"
    from oracle import stage_ref                                                  # staged copies (oracle/stage_ref.py), unpacked to a temp dir
    return stage_ref.unpack() or "/root/reference"


REFERENCE_ROOT = _reference_root()
PAINTER_DIR = os.path.join(REFERENCE_ROOT, "Painter")
SEGGPT_DIR = os.path.join(REFERENCE_ROOT, "SegGPT", "SegGPT_inference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(PAINTER_DIR, "models_painter.py"))


class _Mlp(nn.Module):
    """timm==0.3.2 timm/models/vision_transformer.py Mlp (third-party, restated)."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)

    def forward(self, x):
        x = self.fc1(x)
        x = self.act(x)
        x = self.drop(x)
        x = self.fc2(x)
        x = self.drop(x)
        return x


class _DropPath(nn.Module):
    """timm==0.3.2 timm/models/layers/drop.py DropPath (third-party, restated)."""

    def __init__(self, drop_prob=None):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep_prob = 1 - self.drop_prob
        shape = (x.shape[0],) + (1,) * (x.ndim - 1)
        random_tensor = keep_prob + torch.rand(shape, dtype=x.dtype, device=x.device)
        random_tensor.floor_()
        return x.div(keep_prob) * random_tensor


def _mod(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install_stubs():
    """Inject the absent third-party modules (idempotent)."""
    if "timm" in sys.modules and getattr(sys.modules["timm"], "_painter_stub", False):
        return
    timm = _mod("timm", __version__="0.3.2", _painter_stub=True)
    timm.models = _mod("timm.models")
    timm.models.layers = _mod("timm.models.layers", DropPath=_DropPath,
                              trunc_normal_=torch.nn.init.trunc_normal_)
    timm.models.vision_transformer = _mod("timm.models.vision_transformer", Mlp=_Mlp)

    fv = _mod("fvcore")
    fv.nn = _mod("fvcore.nn")
    fv.nn.weight_init = _mod("fvcore.nn.weight_init",
                             c2_msra_fill=lambda m: None, c2_xavier_fill=lambda m: None)

    class _CNNBlockBase(nn.Module):
        def __init__(self, in_channels, out_channels, stride):
            super().__init__()

    d2 = _mod("detectron2")
    d2.layers = _mod("detectron2.layers", CNNBlockBase=_CNNBlockBase, Conv2d=nn.Conv2d,
                     get_norm=lambda norm, ch: nn.Identity())
    fs = _mod("fairscale")
    fs.nn = _mod("fairscale.nn")
    fs.nn.checkpoint = _mod("fairscale.nn.checkpoint", checkpoint_wrapper=lambda m: m)

    import math
    _mod("torch._six", inf=math.inf)
    if "wandb" not in sys.modules:
        _mod("wandb")
    if "cv2" not in sys.modules:
        _mod("cv2")


def _load(private_name, path, pkg_dir):
    """Import `path` as `private_name`; its `from util.vitdet_utils import ...` resolves to the
    reference's own util package (identical in both trees)."""
    if private_name in sys.modules:
        return sys.modules[private_name]
    install_stubs()
    saved_util = {k: v for k, v in sys.modules.items() if k == "util" or k.startswith("util.")}
    for k in saved_util:
        del sys.modules[k]
    sys.path.insert(0, pkg_dir)
    try:
        spec = importlib.util.spec_from_file_location(private_name, path)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[private_name] = mod
        spec.loader.exec_module(mod)
    finally:
        sys.path.remove(pkg_dir)
        # keep the reference's util.* importable only through the loaded module's globals
        for k in [k for k in sys.modules if k == "util" or k.startswith("util.")]:
            sys.modules["_ref_" + private_name + "." + k] = sys.modules.pop(k)
        sys.modules.update(saved_util)
    return mod


def load_reference_painter():
    """-> the reference module object of Painter/models_painter.py."""
    return _load("ref_models_painter", os.path.join(PAINTER_DIR, "models_painter.py"), PAINTER_DIR)


def load_reference_seggpt():
    """-> the reference module object of SegGPT/SegGPT_inference/models_seggpt.py."""
    return _load("ref_models_seggpt", os.path.join(SEGGPT_DIR, "models_seggpt.py"), SEGGPT_DIR)


def load_reference_engine_train():
    """-> the reference module object of Painter/engine_train.py (train_one_epoch, evaluate_pt) together with its own util.misc /
    util.lr_sched (reachable as module attributes `misc`, `lr_sched`); `wandb` and `torch._six` are stand-ins.  The module only
    touches the MODEL through its public surface, so any module with that surface -- the reference class or
    painter_amd.models_painter.Painter -- can be driven by it unchanged."""
    return _load("ref_engine_train", os.path.join(PAINTER_DIR, "engine_train.py"), PAINTER_DIR)


def load_reference_seggpt_engine():
    """-> the reference module object of SegGPT/SegGPT_inference/seggpt_engine.py (cv2 is a stand-in: only inference_video needs it)."""
    return _load("ref_seggpt_engine", os.path.join(SEGGPT_DIR, "seggpt_engine.py"), SEGGPT_DIR)


def load_reference_pairdataset():
    """-> the reference module object of Painter/data/pairdataset.py.  It needs torchvision only for two base classes
    (`VisionDataset`, `StandardTransform`, pairdataset.py:19); torchvision is absent from this image, so trivial stand-ins that keep
    the constructor arguments are injected.  The transform objects handed to `PairDataset` by the tests are stand-ins as well."""
    install_stubs()
    if "torchvision" not in sys.modules:
        class VisionDataset(torch.utils.data.Dataset):
            def __init__(self, root=None, transforms=None, transform=None, target_transform=None):
                self.root = root
                self.transform = transform
                self.target_transform = target_transform
                self.transforms = transforms

        class StandardTransform:
            def __init__(self, transform=None, target_transform=None):
                self.transform = transform
                self.target_transform = target_transform

        tv = _mod("torchvision", _painter_stub=True)
        tv.datasets = _mod("torchvision.datasets")
        tv.datasets.vision = _mod("torchvision.datasets.vision", VisionDataset=VisionDataset, StandardTransform=StandardTransform)
    return _load("ref_pairdataset", os.path.join(PAINTER_DIR, "data", "pairdataset.py"), os.path.join(PAINTER_DIR, "data"))


def load_reference_masking_generator():
    """-> the reference module object of Painter/util/masking_generator.py (pure Python, no third-party imports)."""
    return _load("ref_masking_generator", os.path.join(PAINTER_DIR, "util", "masking_generator.py"), os.path.join(PAINTER_DIR, "util"))
