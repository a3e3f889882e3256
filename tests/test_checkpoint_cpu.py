"""SURVEY.md 8f row N4 -- checkpoint I/O compatibility, host logic only (no GPU, no kernels).

The reference keeps checkpoints as plain `torch.save` dictionaries of `state_dict()`s (util/misc.py:296-330) and loads released /
MAE weights with `load_state_dict(..., strict=False)` after pruning shape-mismatched keys (main_train.py:198-221,
SegGPT_inference/seggpt_inference.py:40-48).  painter_amd ships no checkpoint code of its own: the UNMODIFIED reference functions are
run here against painter_amd's module, optimizer and loss scaler, and reference-class checkpoints are loaded into painter_amd's classes
and back."""
import importlib.util
import os
import types
from functools import partial

import pytest
import torch
import torch.nn as nn

from oracle import painter_oracle as O
from oracle import ref_import

pytestmark = pytest.mark.skipif(not ref_import.reference_available(), reason="needs /root/reference (build container only)")


def _misc():
    ref_import.install_stubs()
    spec = importlib.util.spec_from_file_location("ref_util_misc", os.path.join(ref_import.PAINTER_DIR, "util", "misc.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _ours(cfg, seggpt=False):
    from painter_amd import models_painter, models_seggpt
    cls = models_seggpt.SegGPT if seggpt else models_painter.Painter
    return cls(img_size=cfg.img_size, patch_size=cfg.patch_size, embed_dim=cfg.embed_dim, depth=cfg.depth, num_heads=cfg.num_heads,
               drop_path_rate=0.1, mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6), use_rel_pos=True,
               decoder_embed_dim=cfg.decoder_embed_dim)


def _reference(cfg):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    try:
        import make_golden
    finally:
        sys.path.pop(0)
    return make_golden.build_reference(cfg, 5)[0]


def test_reference_save_model_and_load_model_round_trip_our_objects(tmp_path):
    misc = _misc()
    from painter_amd import optim as PO
    cfg = O.small_config()
    model = _ours(cfg)
    groups = [{"params": [p for p in model.parameters() if p.ndim > 1], "weight_decay": 0.05, "lr_scale": 0.5},
              {"params": [p for p in model.parameters() if p.ndim <= 1], "weight_decay": 0.0, "lr_scale": 1.0}]
    opt = PO.AdamW(groups, lr=1e-3, betas=(0.9, 0.999))
    for g in opt.param_groups:                                     # state as a few steps would leave it (host tensors here)
        for p in g["params"]:
            opt.state[p] = {"step": torch.tensor(3.0), "exp_avg": torch.randn_like(p), "exp_avg_sq": torch.rand_like(p)}
    scaler = PO.NativeScalerWithGradNormCount()
    args = types.SimpleNamespace(output_dir=str(tmp_path), resume="", start_epoch=0)
    misc.save_model(args=args, epoch=4, model=model, model_without_ddp=model, optimizer=opt, loss_scaler=scaler)
    path = tmp_path / "checkpoint-4.pth"
    assert path.exists()
    ck = torch.load(path, map_location="cpu", weights_only=False)
    assert set(ck) == {"model", "optimizer", "epoch", "scaler", "args"} and ck["epoch"] == 4
    assert list(ck["model"].keys()) == list(O.param_shapes(cfg).keys())                  # the reference's key order and names

    model2 = _ours(cfg)
    groups2 = [{"params": [p for p in model2.parameters() if p.ndim > 1], "weight_decay": 0.05, "lr_scale": 0.5},
               {"params": [p for p in model2.parameters() if p.ndim <= 1], "weight_decay": 0.0, "lr_scale": 1.0}]
    opt2 = PO.AdamW(groups2, lr=1e-3, betas=(0.9, 0.999))
    args2 = types.SimpleNamespace(output_dir=str(tmp_path), resume=str(path), start_epoch=0)
    _orig = torch.load
    torch.load = partial(_orig, weights_only=False)                # the reference predates torch 2.6's weights_only default
    try:
        misc.load_model(args=args2, model_without_ddp=model2, optimizer=opt2, loss_scaler=PO.NativeScalerWithGradNormCount())
    finally:
        torch.load = _orig
    assert args2.start_epoch == 5
    for (k, a), (_, b) in zip(model.state_dict().items(), model2.state_dict().items()):
        assert torch.equal(a, b), k
    for pa, pb in zip(model.parameters(), model2.parameters()):
        for key in ("step", "exp_avg", "exp_avg_sq"):
            assert torch.equal(opt.state[pa][key], opt2.state[pb][key])
    assert [g["lr_scale"] for g in opt2.param_groups] == [0.5, 1.0]


@pytest.mark.parametrize("seggpt", [False, True])
def test_reference_class_checkpoints_load_into_our_classes_and_back(tmp_path, seggpt):
    cfg = O.small_config(seggpt=seggpt)
    ref = _reference(cfg)
    torch.save({"model": ref.state_dict()}, tmp_path / "ref.pth")
    ours = _ours(cfg, seggpt)
    # SegGPT_inference/seggpt_inference.py:44-46 (prepare_model)
    checkpoint = torch.load(tmp_path / "ref.pth", map_location="cpu")
    msg = ours.load_state_dict(checkpoint["model"], strict=False)
    assert not msg.missing_keys and not msg.unexpected_keys
    for (k, a), (k2, b) in zip(ref.state_dict().items(), ours.state_dict().items()):
        assert k == k2 and torch.equal(a, b), k
    ref2 = _reference(cfg)
    with torch.no_grad():
        for p in ref2.parameters():
            p.zero_()
    ref2.load_state_dict(ours.state_dict(), strict=True)
    assert all(torch.equal(a, b) for a, b in zip(ref.state_dict().values(), ref2.state_dict().values()))


def test_mae_fine_tune_load_prunes_mismatched_keys_like_main_train():
    """main_train.py:198-221 restated line by line around our module: an MAE-style checkpoint has no decoder / rel-pos / segment tokens
    and carries a decoder_embed and mask_token of another shape, which are dropped before the strict=False load."""
    cfg = O.small_config()
    model = _ours(cfg)
    sd = model.state_dict()
    enc = {k: torch.randn_like(v) for k, v in sd.items() if k.startswith("blocks.") and "rel_pos" not in k or k.startswith("patch_embed.")}
    checkpoint = {"model": dict(enc)}
    checkpoint["model"]["decoder_embed.weight"] = torch.randn(512, cfg.embed_dim)        # MAE's decoder_embed: another shape
    checkpoint["model"]["decoder_embed.bias"] = torch.randn(512)
    checkpoint["model"]["mask_token"] = torch.randn(1, 1, 512)
    checkpoint["model"]["pos_embed"] = torch.randn_like(sd["pos_embed"])
    checkpoint_model = checkpoint["model"]
    state_dict = model.state_dict()
    rm_key_list = ['decoder_embed.weight', 'decoder_embed.bias', 'mask_token']
    for k in rm_key_list:
        if k in checkpoint_model and checkpoint_model[k].shape != state_dict[k].shape:
            del checkpoint_model[k]
    msg = model.load_state_dict(checkpoint_model, strict=False)
    assert not msg.unexpected_keys
    assert {"decoder_embed.weight", "decoder_embed.bias", "mask_token", "segment_token_x", "segment_token_y"} <= set(msg.missing_keys)
    assert all("rel_pos" in k or not k.startswith("blocks.") for k in msg.missing_keys)
    for k, v in enc.items():
        assert torch.equal(model.state_dict()[k], v), k
