"""SURVEY.md 8a row a18 -- construction-time initialisation (Painter/models_painter.py:320, :337-349): a freshly constructed
painter_amd module must carry the reference's parameter DISTRIBUTIONS (not its RNG stream): zero rel_pos tables
(rel_pos_zero_init), LayerNorm (1, 0), trunc-normal(0.02) Linear weights with zero biases, normal(0.02) tokens, trunc-normal(0.02)
pos_embed, and torch's default Conv2d init for the three conv layers (the reference's _init_weights does not touch them).
Checked against the stated distributions always, and against a live instance of the unmodified reference class when
/root/reference is mounted (build container)."""
import math
from functools import partial

import pytest
import torch
import torch.nn as nn

from oracle import ref_import

KW = dict(img_size=(128, 64), patch_size=16, embed_dim=128, depth=24, num_heads=2, drop_path_rate=0.1, window_size=14, qkv_bias=True,
          mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6), window_block_indexes=([0, 1], [3, 4]), residual_block_indexes=[],
          use_rel_pos=True, out_feature="last_feat", decoder_embed_dim=64, loss_func="smoothl1")


def _ours(seed):
    from painter_amd import models_painter
    torch.manual_seed(seed)
    return models_painter.Painter(**KW)


def _kind(name, p):
    if name.endswith("rel_pos_h") or name.endswith("rel_pos_w"):
        return "zeros"
    if name.endswith(("norm1.weight", "norm2.weight")) or name in ("norm.weight", "decoder_pred.1.weight"):
        return "ones"
    if name.endswith(("norm1.bias", "norm2.bias")) or name in ("norm.bias", "decoder_pred.1.bias"):
        return "zeros"
    if name in ("mask_token", "segment_token_x", "segment_token_y", "type_token_cls", "type_token_ins"):
        return "normal02"
    if name == "pos_embed":
        return "normal02"
    if name.startswith(("patch_embed.proj", "decoder_pred.0", "decoder_pred.3")):
        return "conv_default"
    if name.endswith(".weight"):
        return "normal02"                      # nn.Linear: trunc_normal_(std=.02) (the +-2 cut is 100 sigma away)
    if name.endswith(".bias"):
        return "zeros"                         # nn.Linear bias: constant 0
    raise AssertionError("unclassified parameter " + name)


def _hi(t):
    """|t| at the 1 - 1e-3 quantile (the maximum for small tensors)."""
    v = t.abs().reshape(-1)
    k = max(1, int(v.numel() * (1 - 1e-3)))
    return float(v.kthvalue(k).values)


def test_fresh_module_has_the_reference_initial_distributions():
    m = _ours(0)
    sd = dict(m.named_parameters())
    pooled = {"normal02": []}
    for name, p in sd.items():
        k = _kind(name, p)
        t = p.detach().double()
        if k == "zeros":
            assert float(t.abs().max()) == 0.0, name
        elif k == "ones":
            assert float((t - 1).abs().max()) == 0.0, name
        elif k == "normal02":
            pooled["normal02"].append(t.reshape(-1))
            assert _hi(t) < 0.02 * 4.5, name        # (not the max: trunc_normal_ returns its +-2 clamp about once per 2^24 draws)
            if t.numel() >= 16384:
                assert abs(float(t.std()) / 0.02 - 1) < 0.03 and abs(float(t.mean())) < 0.02 * 4 / math.sqrt(t.numel()) + 1e-9, name
        else:                                  # torch default Conv2d init: kaiming_uniform(a = sqrt 5) = U(-1/sqrt(fan_in), +1/sqrt(fan_in)), bias likewise
            w = sd[name.rsplit(".", 1)[0] + ".weight"]
            bound = 1.0 / math.sqrt(w.shape[1] * w.shape[2] * w.shape[3])
            assert float(t.abs().max()) <= bound * (1 + 1e-6), name
            if t.numel() >= 4096:
                assert abs(float(t.std()) / (bound / math.sqrt(3)) - 1) < 0.05, name
    allv = torch.cat(pooled["normal02"])
    assert abs(float(allv.std()) / 0.02 - 1) < 5e-3 and abs(float(allv.mean())) < 1e-4
    # a second construction draws different values (nothing is a baked constant) with the same constants
    m2 = _ours(1)
    assert not torch.equal(m.blocks[3].attn.qkv.weight, m2.blocks[3].attn.qkv.weight)
    assert torch.equal(m.blocks[3].attn.rel_pos_h, m2.blocks[3].attn.rel_pos_h)


@pytest.mark.skipif(not ref_import.reference_available(), reason="/root/reference not mounted")
def test_fresh_module_statistics_match_a_fresh_reference_module():
    """Same constructor arguments through the UNMODIFIED reference class: per parameter, either both are the same constant, or mean /
    std / extrema agree to sampling error."""
    ref = ref_import.load_reference_painter().Painter
    torch.manual_seed(0)
    r = ref(**KW)
    o = _ours(1)
    rs, os_ = dict(r.named_parameters()), dict(o.named_parameters())
    assert list(rs) == list(os_)
    for name in rs:
        a, b = rs[name].detach().double(), os_[name].detach().double()
        assert a.shape == b.shape, name
        ca, cb = float(a.std()) == 0.0 if a.numel() > 1 else None, float(b.std()) == 0.0 if b.numel() > 1 else None
        assert ca == cb, name
        if ca:
            assert torch.equal(a, b), name
            continue
        n = a.numel()
        if n >= 4096:
            assert abs(float(a.std()) / float(b.std()) - 1) < 6.0 / math.sqrt(n) + 0.02, (name, float(a.std()), float(b.std()))
            assert abs(float(a.mean()) - float(b.mean())) < 6.0 * float(a.std()) / math.sqrt(n), name
        # extremes through a high quantile, not the max: torch's trunc_normal_ (inverse-CDF sampling, clamped to its absolute [-2, 2])
        # returns the clamp value itself about once per 2^24 draws -- in the reference as here, it is the same function
        qa, qb = _hi(a), _hi(b)
        f = 1.3 if n >= 4096 else 3.0             # small tensors: the maximum of ~100 draws is itself a wide random variable
        assert qb <= f * qa + 1e-12 and qa <= f * qb + 1e-12, (name, qa, qb)
