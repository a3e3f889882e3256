"""CPU-side tests (no GPU): the C-ABI library loads and exports every symbol include/painter_hip.h declares, the host
logic of the drop-in modules (constructor surface, parameter names = checkpoint ABI, patchify index math, constant
operators), and that the product path refuses to run without a GPU instead of falling back."""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import painter_oracle as O
from painter_amd import hostmath
from painter_amd._lib import HEADER, LIB_PATH, lib, parse_header


def test_c_abi_library_loads_and_exports_every_declared_symbol():
    import __graft_entry__ as ge
    if not os.path.exists(LIB_PATH):
        ge.build()
    protos = parse_header(HEADER)
    assert len(protos) >= 40
    dll = ctypes.CDLL(LIB_PATH)
    missing = [n for n in protos if not hasattr(dll, n)]
    assert not missing, missing
    assert lib.pa_abi_version() >= 1
    # host-only helpers are callable without a device
    assert lib.pa_relpos_rows_padded(56, 28) == 192
    assert lib.pa_colsum_workspace_bytes(12544, 1024) > 0
    assert lib.pa_linear_wgrad_workspace_bytes(1, 12544, 1024, 1024) > 0
    assert lib.pa_attn_bwd_aux_bytes(8, 1568, 16, 56, 28) > 0


def test_abs_pos_operator_is_the_bicubic_resize():
    m = hostmath.abs_pos_operator(14, 56, 28)
    assert m.shape == (1568, 196)
    np.testing.assert_allclose(m.sum(1), 1.0, atol=1e-5)
    eye = torch.eye(196).reshape(1, 196, 14, 14)
    ref = torch.nn.functional.interpolate(eye, size=(56, 28), mode="bicubic", align_corners=False).reshape(196, 1568).t()
    np.testing.assert_allclose(m, ref.numpy(), atol=2e-6)
    assert np.array_equal(hostmath.abs_pos_operator(14, 14, 14), np.eye(196, dtype=np.float32))


def test_drop_path_rates_match_linspace():
    r = hostmath.drop_path_rates(0.1, 24)
    np.testing.assert_allclose(r, torch.linspace(0, 0.1, 24).numpy(), rtol=0, atol=2e-8)     # float32 ulp
    assert r[0] == 0.0


def test_module_surface_and_checkpoint_abi():
    from painter_amd import models_painter, models_seggpt
    m = models_painter.painter_vit_large_patch16_input896x448_win_dec64_8glb_sl1()
    assert models_painter.PainterViT is models_painter.Painter
    assert m.patch_size == 16 and m.patch_embed.num_patches == 1568 and len(m.blocks) == 24
    assert m.no_weight_decay() == {"pos_embed", "cls_token"}
    cfg = O.vit_large_config()
    ref = O.random_params(cfg, 1)
    sd = m.state_dict()
    assert set(sd.keys()) == set(ref.keys())
    for k, v in ref.items():
        assert tuple(sd[k].shape) == tuple(v.shape), k
    assert sum(p.numel() for p in m.parameters()) == sum(v.numel() for v in ref.values())
    # 1-D parameters stay 1-D (lr_decay's no-weight-decay rule, util/lr_decay.py:32)
    assert sum(1 for p in m.parameters() if p.ndim == 1) == 200
    s = models_seggpt.seggpt_vit_large_patch16_input896x448()
    assert {"type_token_cls", "type_token_ins"} <= set(s.state_dict().keys())
    s.seg_type = "instance"          # set from outside by the caller (SegGPT_inference/seggpt_inference.py:43)
    assert s.seg_type == "instance"


def test_drop_path_factors_come_from_one_draw_and_keep_timm_semantics():
    """timm 0.3.2 drop_path (per sample: floor(keep + U[0,1)) / keep, independent per branch, models_painter.py:255-262 via Block):
    the module draws every block's factors at once; block 0 (rate 0) gets none, blocks after the stream merge are batch-wide, values
    are 0 or 1 / keep, the drop frequency follows the rate, eval mode draws nothing."""
    from painter_amd import models_painter
    m = models_painter.painter_vit_large_patch16_input896x448_win_dec64_8glb_sl1()
    m.eval()
    assert m._drop_scales(4, torch.device("cpu")) is None
    m.train()
    torch.manual_seed(0)
    ds = m._drop_scales(512, torch.device("cpu"))
    assert len(ds) == 24 and ds[0] == (None, None)
    merge = m._cfg.merge_idx
    for i in range(1, 24):
        a, b = ds[i]
        width = 1024 if i <= merge else 512
        assert a.shape == (width,) and b.shape == (width,) and a.is_contiguous() and b.is_contiguous()
        keep = 1.0 - m.blocks[i].drop_path_prob
        for t in (a, b):
            vals = torch.unique(t)
            assert all(abs(float(v)) < 1e-6 or abs(float(v) - 1.0 / keep) < 1e-5 for v in vals)
        assert not torch.equal(a, b)                                   # the two branches draw independently
    dropped = float((ds[23][0] == 0).float().mean())
    assert abs(dropped - m.blocks[23].drop_path_prob) < 0.04          # rate 0.1 at the last block, 512 samples
    assert "_drop_keep" not in m.state_dict()


def test_patchify_roundtrip_bit_exact_and_order():
    from painter_amd import models_painter
    m = models_painter.Painter(img_size=(128, 64), patch_size=16, embed_dim=64, depth=24, num_heads=1, decoder_embed_dim=64,
                               use_rel_pos=True)
    x = torch.arange(2 * 3 * 128 * 64, dtype=torch.float32).reshape(2, 3, 128, 64)
    p = m.patchify(x)
    assert p.shape == (2, 32, 768)
    assert torch.equal(m.unpatchify(p), x)
    assert torch.equal(p, O.patchify(x, 16))
    # patchify(img)[n, l, (p*16+q)*3+c] == img[n, c, h*16+p, w*16+q]   
    assert p[1, 5, (3 * 16 + 7) * 3 + 2] == x[1, 2, (5 // 4) * 16 + 3, (5 % 4) * 16 + 7]
    with pytest.raises(AssertionError):
        m.patchify(torch.zeros(1, 3, 96, 64))


def test_no_cpu_fallback():
    from painter_amd import models_painter
    m = models_painter.Painter(img_size=(128, 64), patch_size=16, embed_dim=64, depth=24, num_heads=1, decoder_embed_dim=64,
                               use_rel_pos=True)
    x = torch.zeros(1, 3, 128, 64)
    with pytest.raises(Exception):
        m(x, x, bool_masked_pos=torch.zeros(1, 32), valid=torch.ones_like(x))


REF_PAINTER = "/root/reference/Painter"


@pytest.mark.skipif(not os.path.isdir(REF_PAINTER), reason="reference tree not mounted (GPU box)")
def test_fused_adamw_takes_the_references_layer_decay_groups_and_lr_schedule():
    """The unchanged reference helpers drive our optimizer object: util/lr_decay.param_groups_lrd (main_train.py:344-347) builds the
    groups, util/lr_sched.adjust_learning_rate (engine_train.py:56) rewrites their lr through `lr_scale`.  Host logic only."""
    import importlib.util
    import types

    def load(name):
        spec = importlib.util.spec_from_file_location("ref_" + name, os.path.join(REF_PAINTER, "util", name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod

    lrd, lr_sched = load("lr_decay"), load("lr_sched")
    from painter_amd import models_painter, optim as PO
    m = models_painter.painter_vit_large_patch16_input896x448_win_dec64_8glb_sl1()
    groups = lrd.param_groups_lrd(m, 0.05, no_weight_decay_list=m.no_weight_decay(), layer_decay=0.8)
    assert len(groups) == 52 and len(groups) <= PO.MAX_GROUPS
    assert sum(len(g["params"]) for g in groups) == len(list(m.parameters()))
    opt = PO.AdamW(groups, lr=1e-3, betas=(0.9, 0.999))
    args = types.SimpleNamespace(lr=1e-3, min_lr=1e-5, warmup_epochs=1, epochs=15)
    lr = lr_sched.adjust_learning_rate(opt, 0.5, args)
    assert abs(lr - 5e-4) < 1e-12
    for g in opt.param_groups:
        assert abs(g["lr"] - lr * g["lr_scale"]) < 1e-15 and g["weight_decay"] in (0.0, 0.05)
    # decay rule of the reference: 1-D parameters and pos_embed are not decayed
    nd = {id(p) for g in opt.param_groups if g["weight_decay"] == 0.0 for p in g["params"]}
    for n, p in m.named_parameters():
        assert (id(p) in nd) == (p.ndim == 1 or n in ("pos_embed", "cls_token")), n
    sd = opt.state_dict()
    assert len(sd["param_groups"]) == 52


def test_sparse_rows_is_a_lossless_form_of_the_position_operator():
    import numpy as np
    for src, h, w in ((14, 56, 28), (16, 64, 32), (14, 8, 4)):
        m = hostmath.abs_pos_operator(src, h, w)
        for mat in (m, m.T):
            idx, val, K = hostmath.sparse_rows(mat)
            assert idx.dtype == np.int32 and val.dtype == np.float32 and idx.shape == val.shape == (mat.shape[0], K)
            back = np.zeros_like(mat)
            for r in range(mat.shape[0]):
                np.add.at(back[r], idx[r], val[r])
            assert np.array_equal(back, mat)
        assert hostmath.sparse_rows(m)[2] <= 16                      # 4 x 4 bicubic taps per output token


def test_feature_taps_default_to_the_references_list_and_other_depths_must_name_theirs():
    """models_painter.py:416 hard-codes the taps [5, 11, 17, 23]; the constructor keeps that default (depth 24: every reference factory) and
    refuses a depth it does not fit unless `feature_taps` names a schedule -- the ViT-H/14 entry point passes depth/4*k - 1."""
    from painter_amd import models_painter
    kw = dict(img_size=(128, 64), patch_size=16, embed_dim=64, num_heads=1, decoder_embed_dim=64, use_rel_pos=True)
    assert models_painter.Painter(depth=24, **kw)._cfg.taps == [5, 11, 17, 23]
    with pytest.raises(NotImplementedError, match="feature_taps"):
        models_painter.Painter(depth=16, **kw)
    with pytest.raises(NotImplementedError, match="feature_taps"):
        models_painter.Painter(depth=32, **kw)                      # the reference list would leave blocks 24-31 without gradient
    assert models_painter.Painter(depth=16, feature_taps=(3, 7, 11, 15), **kw)._cfg.taps == [3, 7, 11, 15]
    import inspect
    src = inspect.getsource(models_painter.painter_vit_huge_patch14_input896x448)
    assert "feature_taps=(7, 15, 23, 31)" in src
