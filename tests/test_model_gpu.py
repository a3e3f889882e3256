"""Whole-model GPU parity: the HIP path (through the drop-in nn.Module and the C ABI) against
  (1) golden vectors produced by the UNMODIFIED reference (tests/golden/*.npz) and
  (2) the CPU oracle on the same seeded inputs.
fp32 build: the north-star gate, 1e-3 relative (we assert tighter); bf16 build: loss to 2e-3, pred against the
reference's own bf16-autocast deviation (~1e-2)."""
from functools import partial

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import painter_oracle as O
from tests import golden_util as G

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from painter_amd import models_painter, models_seggpt


def build(cfg, seed, dtype, train=False):
    cls = models_seggpt.SegGPT if cfg.seggpt else models_painter.Painter
    m = cls(img_size=cfg.img_size, patch_size=cfg.patch_size, embed_dim=cfg.embed_dim, depth=cfg.depth, num_heads=cfg.num_heads,
            drop_path_rate=0.1, window_size=14, qkv_bias=True, mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6),
            window_block_indexes=([0, 1], [3, 4]), residual_block_indexes=[], use_rel_pos=True, out_feature="last_feat",
            decoder_embed_dim=cfg.decoder_embed_dim, loss_func=cfg.loss_func, compute_dtype=dtype,
            **({} if cfg.depth == 24 else {"feature_taps": cfg.taps}))       # depth 24: the reference's own call, its hard-coded taps
    assert tuple(m._cfg.taps) == tuple(cfg.taps) and m._cfg.merge_idx == cfg.merge_idx
    P = O.random_params(cfg, seed)
    missing = m.load_state_dict(P, strict=True)
    m = m.cuda()
    m.train(train)
    return m, P


# Gates on the bf16 build's gradients against the unmodified reference's fp32 gradients (the tests print what they measure, pytest -s).
# Since round 5 they are tied to a MEASUREMENT OF THE REFERENCE ITSELF (tools/grad_yardstick.py on the GPU box,
# profiles/r05_reference_bf16_gradient_yardstick.json): the unmodified reference Painter at ViT-L, B = 1, under
# torch.autocast(bfloat16) -- the arrangement of engine_train.py:65-75 -- against its own fp32 run, same parameters and batch as
# tests/golden/painter_vitl.npz, same metrics as below:
#                                                        reference bf16 autocast      HIP bf16 build (same process)
#    sampled rel-max, every tensor but the rel-pos tables      1.95e-2                     2.26e-2        (medians 9.8e-3 / 8.3e-3)
#    sampled rel-max, rel-pos tables                           1.28e-1                     1.07e-1
#    rel-pos tables, full-tensor relative Frobenius            4.17e-2                     3.71e-2        (medians 3.9e-2 / 3.3e-2)
# Gates = 1.5 x the reference's own deviation (3.0e-2, 1.9e-1); the Frobenius gate stays at the tighter 5e-2 of round 4 (1.5 x would be
# 6.3e-2).  So the error of the rel-pos table gradients -- the noisiest tensors of the model: class sums of dS = P o (dP - Delta), whose rows
# sum to zero, under a 2^-9 rounding of every P, dS and bias-table entry -- is a property of bf16 arithmetic on this model that the
# reference shows to the same degree, not a loose kernel.  The head_dim-80 small model has short tensors (few samples per tensor, larger
# spread: 4.0e-2 .. 6.3e-2 measured); its yardstick is taken LIVE by tests/test_live_yardstick_gpu.py (the reference's own bf16 autocast on the same
# cases: 3.0e-2 .. 7.3e-2, the HIP build at 0.55 - 0.80 x of it); the fixture tests keep the reference's own worst, 7.3e-2 (round 5: 1e-1).
BF16_SAMPLE_GATE = 2.45e-2         # every tensor except the rel-pos tables, ViT-L fixtures: 1.25 x the reference's own 1.95e-2 (round 6; 1.5 x before)
# Round 6 ("find what blocks.0.attn.qkv.weight loses").  Nothing: the sampled rel-max is the MAXIMUM over ~3000 samples
# of one tensor, a heavy-tailed estimator -- per block and weight family (tools/grad_yardstick.py, profiles/r06_grad_yardstick_per_block.log) the HIP
# bf16 build sits BELOW the reference's own bf16-autocast deviation on every one of the 96 big matrices by whole-tensor relative Frobenius error
# (qkv.weight of block 0: 9.9e-3 against the reference's 1.07e-2; worst matrix 1.12e-2 against 1.28e-2; both grow by ~10 % from block 23 to
# block 0: rounding accumulated along the residual chain, the same in both), and its rel-max exceeds the reference's on one tensor of 101 by
# chance.  So the gate that can see a systematic loss is a Frobenius one: over the stored samples of each of the blocks' 96 weight matrices,
# at 1.0 x the reference's own worst matrix.
BF16_SAMPLE_FRO_GATE = 1.3e-2      # ViT-L fixtures: relative Frobenius error over the samples of a tensor (reference's own worst matrix: 1.28e-2; HIP: 1.12e-2)
BF16_SAMPLE_GATE_SHORT = 7.3e-2    # the head_dim-80 small models: the reference's own bf16-autocast deviation on them
BF16_RELPOS_FRO_GATE = 5.0e-2      # rel-pos tables: relative Frobenius error over the full tensor (reference's own: 4.17e-2)
BF16_RELPOS_GATE = 1.9e-1          # rel-pos tables: sampled rel-max, 1.5 x the reference's own 1.28e-1


def _check_bf16_samples(fx, case, m, tag, rtol_norm=1e-1, atol_dot=5e-2, small_rtol=1e-1, sample_gate=BF16_SAMPLE_GATE, sample_fro_gate=None):
    """bf16 build against a reference fixture: digests at the model-level bf16 bounds, the sampled gradients at the gates above."""
    rep = []
    G.check_grad_digests(fx, case, [(n, p.grad) for n, p in m.named_parameters()], rtol_norm, atol_dot, small_rtol, sample_rtol=1e9, report=rep)
    rel = [r for r in rep if "rel_pos" in r[1]]
    oth = [r for r in rep if "rel_pos" not in r[1]]
    fro = []
    for n, p in m.named_parameters():
        key = "%sgrad_full/%s" % (case, n)
        if key in fx.files:
            fro.append((G.rel_fro(p.grad.detach().float().cpu().reshape(-1), torch.as_tensor(fx[key]).reshape(-1)), n))
    print("%s: worst sampled-gradient rel-max error: rel-pos tables %.3e (%s), all other tensors %.3e (%s), %d tensors; rel-pos tables, "
          "full-tensor rel-Frobenius: worst %.3e (%s) over %d tables"
          % ((tag,) + (max(rel) if rel else (0.0, "-")) + (max(oth) if oth else (0.0, "-")) + (len(rep),) + (max(fro) if fro else (0.0, "-")) + (len(fro),)))
    assert not rel or max(rel)[0] < BF16_RELPOS_GATE, max(rel)
    assert not oth or max(oth)[0] < sample_gate, max(oth)
    assert not fro or max(fro)[0] < BF16_RELPOS_FRO_GATE, max(fro)
    if sample_fro_gate is not None:
        sfro = []
        for n, p in m.named_parameters():
            key = "%sgrad_sample/%s" % (case, n)
            if key in fx.files and n.startswith("blocks.") and n.endswith(".weight") and p.dim() == 2:      # the 96 matrices the yardstick lists per block
                b = torch.as_tensor(fx[key]).reshape(-1)
                a = p.grad.detach().float().cpu().reshape(-1)
                sfro.append((G.rel_fro(a[::int(fx[case + "grad_sample_stride"])], b), n))
        print("%s: worst relative Frobenius error over the samples of a block's weight matrix: %.3e (%s), %d tensors" % ((tag,) + max(sfro) + (len(sfro),)))
        assert max(sfro)[0] < sample_fro_gate, max(sfro)


def run_painter(m, cfg, batch, seed_x, mask_kind, backward=True):
    imgs, tgts, mask, valid = O.synthetic_batch(cfg, batch, seed_x, mask_kind)
    valid_d = valid.cuda()
    for p in m.parameters():
        p.grad = None
    loss, pred, mo = m(imgs.cuda(), tgts.cuda(), bool_masked_pos=mask.reshape(batch, *cfg.grid).cuda(), valid=valid_d)
    if backward:
        loss.backward()
    torch.cuda.synchronize()
    return loss, pred, mo, valid_d


def test_small_fp32_vs_reference_golden_and_oracle():
    fx = G.load("painter_small.npz")
    case, cfg = "painter_rand/", O.small_config()
    m, P = build(cfg, 11, "fp32")
    loss, pred, mo, valid_d = run_painter(m, cfg, 2, 21, "random")
    ref_loss = float(fx[case + "loss"])
    assert abs(loss.item() - ref_loss) < 1e-4 * abs(ref_loss), (loss.item(), ref_loss)
    e = G.rel_err(pred.cpu(), fx[case + "pred"])
    assert e < 2e-4, e
    assert np.array_equal(mo.cpu().numpy(), fx[case + "mask_out"])
    assert valid_d.double().sum().item() == float(fx[case + "valid_out_sum"])
    G.check_grad_digests(fx, case, [(n, p.grad) for n, p in m.named_parameters()], 1e-3, 1e-3, 1e-3)
    # same inputs through the oracle at run time (no fixture): forward + every gradient tensor
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    imgs, tgts, mask, valid = O.synthetic_batch(cfg, 2, 21, "random")
    lo, po, _ = O.forward(Pg, cfg, imgs, tgts, mask, valid)
    lo.backward()
    assert abs(loss.item() - lo.item()) < 1e-4 * abs(lo.item())
    assert G.rel_err(pred.cpu(), po.detach()) < 2e-4
    worst = max((G.rel_fro(p.grad.cpu(), Pg[n].grad), n) for n, p in m.named_parameters())
    assert worst[0] < 1e-3, worst


def test_small_train_mode_droppath_vs_reference_golden():
    """DropPath with the reference's recorded per-sample factors (timm 0.3.2 semantics), fwd + bwd."""
    fx = G.load("painter_small.npz")
    case, cfg = "painter_train/", O.small_config()
    m, _ = build(cfg, 12, "fp32", train=True)
    flat = torch.from_numpy(fx[case + "drop_scales_flat"])
    chunks = list(torch.split(flat, [int(x) for x in fx[case + "drop_scales_len"]]))
    m._drop_override = [(None, None)] + [(chunks[2 * i].cuda().contiguous(), chunks[2 * i + 1].cuda().contiguous()) for i in range(cfg.depth - 1)]
    loss, pred, _, _ = run_painter(m, cfg, 2, 22, "half")
    ref_loss = float(fx[case + "loss"])
    assert abs(loss.item() - ref_loss) < 1e-4 * abs(ref_loss), (loss.item(), ref_loss)
    assert G.rel_err(pred.cpu(), fx[case + "pred"]) < 2e-4
    G.check_grad_digests(fx, case, [(n, p.grad) for n, p in m.named_parameters()], 1e-3, 1e-3, 1e-3)
    # statistical check of the module's own sampler: factors are 0 or 1/keep, never applied to block 0
    m._drop_override = None
    ds = m._drop_scales(64, torch.device("cuda"))
    assert ds[0] == (None, None) and ds[1][0].shape == (128,) and ds[5][0].shape == (64,)
    keep = 1.0 - m.blocks[23].drop_path_prob
    vals = torch.unique(ds[23][0].cpu())
    assert all(abs(v) < 1e-6 or abs(v - 1.0 / keep) < 1e-5 for v in vals.tolist())


def test_small_bf16_vs_reference_golden():
    fx = G.load("painter_small.npz")
    case, cfg = "painter_rand/", O.small_config()
    m, _ = build(cfg, 11, "bf16")
    loss, pred, _, _ = run_painter(m, cfg, 2, 21, "random")
    ref_loss = float(fx[case + "loss"])
    assert abs(loss.item() - ref_loss) < 2e-3 * abs(ref_loss), (loss.item(), ref_loss)
    assert G.rel_fro(pred.cpu(), fx[case + "pred"]) < 3e-2
    G.check_grad_digests(fx, case, [(n, p.grad) for n, p in m.named_parameters()], 8e-2, 5e-2, 8e-2)


@pytest.mark.parametrize("case,n,merge,seg,seed_x", [("seggpt_n3_merge/", 3, 0, "instance", 23), ("seggpt_n1/", 1, -1, "semantic", 24)])
def test_small_seggpt_inference_vs_reference_golden(case, n, merge, seg, seed_x):
    fx = G.load("painter_small.npz")
    cfg = O.small_config(seggpt=True)
    m, _ = build(cfg, 13, "fp32")
    imgs, tgts, _, valid = O.synthetic_batch(cfg, n, seed_x, "half")
    L = cfg.grid[0] * cfg.grid[1]
    mask = torch.zeros(1, L)
    mask[:, L // 2:] = 1
    seg_type = torch.ones(n, 1) if seg == "instance" else torch.zeros(n, 1)
    with torch.no_grad():
        loss, pred, mo = m(imgs.cuda(), tgts.cuda(), mask.cuda(), valid.cuda(), seg_type.cuda(), merge)
    ref_loss = float(fx[case + "loss"])
    assert abs(loss.item() - ref_loss) < 1e-4 * abs(ref_loss), (loss.item(), ref_loss)
    assert G.rel_err(pred.cpu(), fx[case + "pred"]) < 2e-4
    assert mo.shape == (1, L) and mo.dtype == torch.bool


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("n,merge", [(4, 0), (3, 1)])
def test_small_seggpt_feature_ensemble_under_autograd_vs_oracle(dtype, n, merge):
    """The SegGPT feature ensemble (Block.forward(x, merge), models_seggpt.py:207-238: query-half tokens of the attention branch are
    averaged over the prompts, per stream half before the early merge and over all prompts after it) is differentiable in the
    reference, which only ever runs it under no_grad.  Here: eval mode (no DropPath factor between the ensemble and the residual add),
    loss.backward() through the HIP path against the oracle's autograd on the same parameters and batch -- every parameter gradient;
    then train mode with DropPath factors, with and without the ensemble."""
    cfg = O.small_config(seggpt=True)
    m, P = build(cfg, 17, dtype)
    imgs, tgts, _, valid = O.synthetic_batch(cfg, n, 31, "half")
    L = cfg.grid[0] * cfg.grid[1]
    mask = torch.zeros(1, L)
    mask[:, L // 2:] = 1
    seg_type = torch.ones(n, 1)
    seg_type[0] = 0                                      # both segmentation-type tokens get a gradient
    loss, pred, _ = m(imgs.cuda(), tgts.cuda(), mask.cuda(), valid.clone().cuda(), seg_type.cuda(), merge)
    loss.backward()
    Po = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    lo, _, _ = O.forward(Po, cfg, imgs, tgts, mask.bool().expand(n, L), valid.clone(), seg_type, merge)
    lo.backward()
    tol_l, tol_g = (1e-4, 2e-3) if dtype == "fp32" else (3e-3, 1e-1)
    assert abs(loss.item() - lo.item()) < tol_l * abs(lo.item()), (loss.item(), lo.item())
    worst = ("", 0.0)
    for name, p in m.named_parameters():
        go = Po[name].grad
        if p.grad is None or go is None:              # an unused parameter (the other segmentation-type token): None here, None or zeros there
            assert (go is None or float(go.abs().max()) == 0.0) and (p.grad is None or float(p.grad.abs().max()) == 0.0), name
            continue
        den = float(go.abs().max())
        if den == 0.0:
            assert float(p.grad.abs().max()) == 0.0, name
            continue
        e = float((p.grad.cpu().reshape(go.shape) - go).abs().max()) / den
        if e > worst[1]:
            worst = (name, e)
    print("ensemble under autograd, %s, n=%d merge=%d: loss %.6f vs oracle %.6f, worst gradient %s %.2e" % ((dtype, n, merge, loss.item(), lo.item()) + worst))
    assert worst[1] < tol_g, worst
    # train mode: DropPath factors between the ensemble and the residual add (one per block here, shared by its two branches, so that the
    # oracle can take them), with the ensemble (x1 = x0 + s * ens(a)) and without it (merge_between_batch = -1: an ordinary training step
    # of the SegGPT module)
    m.train()
    g = torch.Generator().manual_seed(5)
    scales, over = [], []
    for i, blk in enumerate(m.blocks):
        if blk.drop_path_prob <= 0.0:
            scales.append(None)
            over.append((None, None))
            continue
        keep = 1.0 - blk.drop_path_prob
        bc = 2 * n if i <= cfg.merge_idx else n
        sc = torch.floor(keep + torch.rand(bc, generator=g)) / keep
        if i in (1, cfg.depth - 1):
            sc[i % bc] = 0.0                         # one sample certainly dropped before and after the early merge
        scales.append(sc)
        over.append((sc.cuda(), sc.cuda()))
    assert any(sc is not None and float(sc.min()) == 0.0 for sc in scales)        # some sample really is dropped somewhere
    m._drop_override = over
    for mg in (merge, -1):
        for p_ in m.parameters():
            p_.grad = None
        loss, _, _ = m(imgs.cuda(), tgts.cuda(), mask.cuda(), valid.clone().cuda(), seg_type.cuda(), mg)
        loss.backward()
        Pt = {k: v.clone().requires_grad_(True) for k, v in P.items()}
        lt, _, _ = O.forward(Pt, cfg, imgs, tgts, mask.bool().expand(n, L), valid.clone(), seg_type, mg, drop_scales=scales)
        lt.backward()
        assert abs(loss.item() - lt.item()) < tol_l * abs(lt.item()), (mg, loss.item(), lt.item())
        worst = ("", 0.0)
        for name, p_ in m.named_parameters():
            go = Pt[name].grad
            if p_.grad is None or go is None or float(go.abs().max()) == 0.0:
                continue
            e = float((p_.grad.cpu().reshape(go.shape) - go).abs().max()) / float(go.abs().max())
            if e > worst[1]:
                worst = (name, e)
        print("train mode, merge_between_batch=%d: loss %.6f vs oracle %.6f, worst gradient %s %.2e" % ((mg, loss.item(), lt.item()) + worst))
        assert worst[1] < tol_g, (mg, worst)


def test_ignore_rule_and_determinism():
    cfg = O.small_config()
    m, P = build(cfg, 1, "bf16")
    imgs, tgts, mask, valid = O.synthetic_batch(cfg, 2, 3, "half")
    mean = torch.tensor(O.IMAGENET_MEAN)[None, :, None, None]
    std = torch.tensor(O.IMAGENET_STD)[None, :, None, None]
    tgts[1] = ((torch.zeros(1, 3, 128, 64) - mean) / std)[0]
    v1 = valid.clone().cuda()
    l1, p1, _ = m(imgs.cuda(), tgts.cuda(), mask.cuda(), v1)
    l1.backward()
    g1 = [p.grad.clone() for p in m.parameters()]
    assert v1[0].min() == 1.0 and v1[1].max() == 0.0                 # in-place mutation, models_painter.py:448
    vo = valid.clone()
    lo, _, _ = O.forward(P, cfg, imgs, tgts, mask, vo)
    assert abs(l1.item() - lo.item()) < 3e-3 * abs(lo.item())
    for p in m.parameters():
        p.grad = None
    v2 = valid.clone().cuda()
    l2, p2, _ = m(imgs.cuda(), tgts.cuda(), mask.cuda(), v2)
    l2.backward()
    assert torch.equal(l1, l2) and torch.equal(p1, p2)                # same input twice -> bit-identical
    assert all(torch.equal(a, p.grad) for a, p in zip(g1, m.parameters()))


def test_vit_large_fp32_vs_reference_golden():
    """BASELINE configs[0]/[1] shape (896x448, ViT-L), B=1, against the reference's own CPU fp32 output."""
    fx = G.load("painter_vitl.npz")
    case, cfg = "vitl_b1/", O.vit_large_config()
    m, _ = build(cfg, 1, "fp32")
    loss, pred, _, _ = run_painter(m, cfg, 1, 1234, "random")
    ref_loss = float(fx[case + "loss"])
    assert abs(loss.item() - ref_loss) < 1e-4 * abs(ref_loss), (loss.item(), ref_loss)
    flat = pred.reshape(-1).cpu()
    stride = int(fx[case + "pred_stride"])
    assert G.rel_err(flat[::stride], fx[case + "pred_sample"]) < 1e-3
    assert abs(float(flat.double().norm()) - float(fx[case + "pred_norm"])) < 1e-4 * float(fx[case + "pred_norm"])
    rep = []
    G.check_grad_digests(fx, case, [(n, p.grad) for n, p in m.named_parameters()], 1e-3, 1e-3, 1e-3, sample_rtol=1e-3, report=rep)
    print("ViT-L B=1 fp32: worst sampled-gradient rel-max error %.3e (%s) over %d tensors" % (max(rep) + (len(rep),)))


def test_vit_large_bf16_loss_and_pred_yardstick():
    fx = G.load("painter_vitl.npz")
    case, cfg = "vitl_b1/", O.vit_large_config()
    m, _ = build(cfg, 1, "bf16")
    loss, pred, _, _ = run_painter(m, cfg, 1, 1234, "random")
    ref_loss = float(fx[case + "loss"])
    assert abs(loss.item() - ref_loss) < 2e-3 * abs(ref_loss), (loss.item(), ref_loss)
    flat = pred.reshape(-1).cpu()
    stride = int(fx[case + "pred_stride"])
    assert G.rel_fro(flat[::stride], fx[case + "pred_sample"]) < 3e-2      # reference's own bf16 deviation: 1e-2
    _check_bf16_samples(fx, case, m, "ViT-L B=1 bf16", sample_fro_gate=BF16_SAMPLE_FRO_GATE)


def _vitl_b8_case(dtype):
    fx = G.load("painter_vitl_b8.npz")
    case, cfg = "vitl_b8_train/", O.vit_large_config()
    m, _ = build(cfg, 1, dtype, train=True)
    flat = torch.from_numpy(fx[case + "drop_scales_flat"])
    chunks = list(torch.split(flat, [int(x) for x in fx[case + "drop_scales_len"]]))
    assert len(chunks) == 2 * (cfg.depth - 1) and chunks[0].numel() == 16 and chunks[-1].numel() == 8
    m._drop_override = [(None, None)] + [(chunks[2 * i].cuda().contiguous(), chunks[2 * i + 1].cuda().contiguous()) for i in range(cfg.depth - 1)]
    loss, pred, _, _ = run_painter(m, cfg, 8, 4321, "random")
    stride = int(fx[case + "pred_stride"])
    ps = pred.reshape(8, -1)[:, ::stride].cpu()
    return fx, case, m, loss, ps


def test_vit_large_b8_train_fp32_vs_reference_golden():
    """BASELINE configs[1] at its real batch (B = 8), TRAIN mode with the reference's recorded DropPath factors, exact-fp32 build:
    loss, every sample's pred, every parameter gradient against the fixture assembled from eight B = 1 runs of the unmodified
    reference (tests/golden/make_golden.py::case_vit_large_b8_train) -- also pins that assembly."""
    fx, case, m, loss, ps = _vitl_b8_case("fp32")
    ref_loss = float(fx[case + "loss"])
    assert abs(loss.item() - ref_loss) < 1e-4 * abs(ref_loss), (loss.item(), ref_loss)
    for b_ in range(8):
        assert G.rel_err(ps[b_], fx[case + "pred_sample"][b_]) < 1e-3, b_
    rep = []
    G.check_grad_digests(fx, case, [(n, p.grad) for n, p in m.named_parameters()], 1e-3, 1e-3, 1e-3, sample_rtol=1e-3, report=rep)
    print("ViT-L B=8 train fp32: worst sampled-gradient rel-max error %.3e (%s) over %d tensors" % (max(rep) + (len(rep),)))


def test_vit_large_b8_train_bf16_vs_reference_golden():
    """The timed configuration itself (bf16 operands, B = 8, train mode, the kernels bench.py runs) against the reference fixture.
    Tolerances: loss 2e-3; pred relative Frobenius 3e-2 per sample (the reference's own bf16-autocast deviation is 1e-2,
    BASELINE.md section 4); gradient digests at the bf16 model-level bound."""
    fx, case, m, loss, ps = _vitl_b8_case("bf16")
    ref_loss = float(fx[case + "loss"])
    assert abs(loss.item() - ref_loss) < 2e-3 * abs(ref_loss), (loss.item(), ref_loss)
    worst = max(G.rel_fro(ps[b_], fx[case + "pred_sample"][b_]) for b_ in range(8))
    assert worst < 3e-2, worst
    _check_bf16_samples(fx, case, m, "ViT-L B=8 train bf16 (the timed configuration)", sample_fro_gate=BF16_SAMPLE_FRO_GATE)


def test_seggpt_vit_large_n32_ensemble_and_hipgraph_vs_reference_golden():
    """BASELINE configs[3]: seggpt_vit_large_patch16_input896x448, 32 prompts over one query, feature ensemble from block 0, bf16,
    forward: eager against the unmodified reference's output (fixture), then captured in a hipGraph and replayed -- the replay must
    be bit-identical to the eager run (same kernels, no host sync inside the forward)."""
    fx = G.load("seggpt_vitl_n32.npz")
    case, cfg, N = "seggpt_n32/", O.vit_large_config(seggpt=True), 32
    m, _ = build(cfg, 2, "bf16")
    imgs, tgts, _, valid = O.synthetic_batch(cfg, N, 777, "half")
    imgs[:, :, cfg.img_size[0] // 2:, :] = imgs[0:1, :, cfg.img_size[0] // 2:, :]
    L = cfg.grid[0] * cfg.grid[1]
    mask = torch.zeros(1, L)
    mask[:, L // 2:] = 1
    imgs, tgts, valid, mask, seg_type = imgs.cuda(), tgts.cuda(), valid.cuda(), mask.cuda(), torch.ones(N, 1).cuda()

    def fwd():
        with torch.no_grad():
            return m(imgs, tgts, mask, valid, seg_type, 0)

    loss, pred, mo = fwd()
    torch.cuda.synchronize()
    ref_loss = float(fx[case + "loss"])
    assert abs(loss.item() - ref_loss) < 3e-3 * abs(ref_loss), (loss.item(), ref_loss)
    stride = int(fx[case + "pred_stride"])
    ps = pred.reshape(N, -1)[:, ::stride].float().cpu()
    worst = max(G.rel_fro(ps[i], fx[case + "pred_sample"][i]) for i in range(N))
    assert worst < 3e-2, worst
    norms = pred.reshape(N, -1).double().norm(dim=1).cpu().numpy()
    assert np.abs(norms / fx[case + "pred_norm"] - 1).max() < 1e-2
    assert mo.shape == (1, L) and mo.dtype == torch.bool
    eager_pred, eager_loss = pred.clone(), loss.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):          # warm-up on the capture stream (per-stream workspaces)
        fwd()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        g_loss, g_pred, _ = fwd()
    g_pred.zero_()
    graph.replay()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(g_pred, eager_pred) and torch.equal(g_loss, eager_loss)


def test_ddp_gradient_path_single_rank_rccl():
    """The multi-GPU gradient exchange exercised on one GPU (tools/ddp_selftest.py): 1-rank RCCL group, GradSync driven from the
    two-stream backward; gradients must be bit-identical to the run without the exchange (this caught a cross-stream allocator
    race on the flattened small-tensor message)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PAINTER_AMD_DDP_SELFTEST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "ddp_selftest.py")], cwd=root, env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "DDP selftest OK" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_three_training_steps_match_cpu_reference_loop():
    """End to end: forward + hand-written backward (fp32 build) + fused AdamW with global-norm clipping, three steps, against the
    same loop on the CPU oracle (autograd of oracle.painter_oracle.forward + torch.optim.AdamW + clip_grad_norm_)."""
    from painter_amd import optim as PO
    cfg = O.small_config()
    m, P = build(cfg, 13, "fp32")
    names = [n for n, _ in m.named_parameters()]
    groups = lambda params: [{"params": [p for n, p in params if p.ndim > 1], "weight_decay": 0.05},
                             {"params": [p for n, p in params if p.ndim <= 1], "weight_decay": 0.0}]
    # eps = 1e-3: with the default 1e-8 the normalised update of an element whose gradient is ~0 is decided by rounding noise
    # (sign flips worth 2 lr), which would test the noise, not the path
    opt = PO.AdamW(groups(list(m.named_parameters())), lr=2e-3, betas=(0.9, 0.95), eps=1e-3).bind_model(m)
    Pc = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    ref = torch.optim.AdamW(groups([(n, Pc[n]) for n in names]), lr=2e-3, betas=(0.9, 0.95), eps=1e-3)
    for it in range(3):
        imgs, tgts, mask, valid = O.synthetic_batch(cfg, 2, 40 + it, "random")
        opt.zero_grad()
        loss, _, _ = m(imgs.cuda(), tgts.cuda(), bool_masked_pos=mask.reshape(2, *cfg.grid).cuda(), valid=valid.clone().cuda())
        loss.backward()
        opt.grad_sumsq()
        opt.step(max_norm=3.0)
        ref.zero_grad()
        lo, _, _ = O.forward(Pc, cfg, imgs, tgts, mask, valid.clone())
        lo.backward()
        torch.nn.utils.clip_grad_norm_([Pc[n] for n in names], 3.0)
        ref.step()
        assert abs(loss.item() - lo.item()) <= 2e-4 * abs(lo.item()), (it, loss.item(), lo.item())
    worst = max(float((p.detach().cpu() - Pc[n].detach()).abs().max() / Pc[n].detach().abs().max().clamp_min(1e-6)) for n, p in m.named_parameters())
    assert worst < 1e-3, worst


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_two_forwards_before_one_backward_and_interleaved_modules(dtype):
    """Autograd bookkeeping of the one-node forward/backward: two forward passes (different inputs) whose losses are summed and
    differentiated once, with a SECOND module's forward + backward run in between, must give the sum of the separately computed
    gradients -- nothing of a call's saved state may live in scratch that a later call reuses.  fp32 build: to rounding; bf16 build:
    bit for bit (same kernels, same order per call)."""
    cfg = O.small_config()
    m, _ = build(cfg, 31, dtype)
    other, _ = build(cfg, 32, dtype)
    batches = [O.synthetic_batch(cfg, 2, 70 + k, "random") for k in range(3)]

    def fwd(mod, k):
        imgs, tgts, mask, valid = batches[k]
        return mod(imgs.cuda(), tgts.cuda(), bool_masked_pos=mask.reshape(2, *cfg.grid).cuda(), valid=valid.cuda())[0]

    def grads(mod):
        return [p.grad.detach().clone() for p in mod.parameters()]

    sep = []
    for k in (0, 1):
        for p in m.parameters():
            p.grad = None
        fwd(m, k).backward()
        sep.append(grads(m))
    for p in m.parameters():
        p.grad = None
    l0 = fwd(m, 0)
    lo = fwd(other, 2)                      # another module's whole step between our two forwards
    lo.backward()
    l1 = fwd(m, 1)
    (l0 + l1).backward()
    torch.cuda.synchronize()
    for g, a, b in zip(grads(m), sep[0], sep[1]):
        ref = a + b
        if dtype == "bf16":
            assert torch.equal(g, ref)
        else:
            assert float((g - ref).abs().max()) <= 1e-6 * float(ref.abs().max().clamp_min(1e-12)) + 1e-12
    assert all(p.grad is not None for p in other.parameters())


# ------------------------------------------------------------------------------------------ ViT-H/14-shaped configurations
def test_h14_fp32_vs_reference_golden_and_oracle():
    """head_dim 80 / patch 14 (BASELINE configs[4]'s arithmetic) at the one depth the UNMODIFIED reference can run it (24: its hard-coded
    taps are the generalised ones there), fixture tests/golden/painter_h14.npz: loss, pred, mask, every gradient; then the oracle at
    run time, every gradient tensor in full."""
    fx = G.load("painter_h14.npz")
    case, cfg = "h14_rand/", O.h14_small_config(depth=24)
    m, P = build(cfg, 31, "fp32")
    loss, pred, mo, valid_d = run_painter(m, cfg, 2, 41, "random")
    ref_loss = float(fx[case + "loss"])
    assert abs(loss.item() - ref_loss) < 1e-4 * abs(ref_loss), (loss.item(), ref_loss)
    assert G.rel_err(pred.cpu(), fx[case + "pred"]) < 2e-4
    assert np.array_equal(mo.cpu().numpy(), fx[case + "mask_out"])
    rep = []
    G.check_grad_digests(fx, case, [(n, p.grad) for n, p in m.named_parameters()], 1e-3, 1e-3, 1e-3, report=rep)
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    imgs, tgts, mask, valid = O.synthetic_batch(cfg, 2, 41, "random")
    lo, po, _ = O.forward(Pg, cfg, imgs, tgts, mask, valid)
    lo.backward()
    worst = max((G.rel_fro(p.grad.cpu(), Pg[n].grad), n) for n, p in m.named_parameters())
    assert worst[0] < 1e-3, worst
    print("h14 fp32: worst sampled-gradient rel-max %.3e (%s); worst full-tensor rel-fro vs oracle %.3e (%s)" % (max(rep) + worst))


def test_h14_train_mode_droppath_vs_reference_golden():
    fx = G.load("painter_h14.npz")
    case, cfg = "h14_train/", O.h14_small_config(depth=24)
    m, _ = build(cfg, 32, "fp32", train=True)
    flat = torch.from_numpy(fx[case + "drop_scales_flat"])
    chunks = list(torch.split(flat, [int(x) for x in fx[case + "drop_scales_len"]]))
    m._drop_override = [(None, None)] + [(chunks[2 * i].cuda().contiguous(), chunks[2 * i + 1].cuda().contiguous()) for i in range(cfg.depth - 1)]
    loss, pred, _, _ = run_painter(m, cfg, 2, 42, "half")
    ref_loss = float(fx[case + "loss"])
    assert abs(loss.item() - ref_loss) < 1e-4 * abs(ref_loss), (loss.item(), ref_loss)
    assert G.rel_err(pred.cpu(), fx[case + "pred"]) < 2e-4
    G.check_grad_digests(fx, case, [(n, p.grad) for n, p in m.named_parameters()], 1e-3, 1e-3, 1e-3)


def test_h14_bf16_vs_reference_golden():
    fx = G.load("painter_h14.npz")
    case, cfg = "h14_rand/", O.h14_small_config(depth=24)
    m, _ = build(cfg, 31, "bf16")
    loss, pred, _, _ = run_painter(m, cfg, 2, 41, "random")
    ref_loss = float(fx[case + "loss"])
    assert abs(loss.item() - ref_loss) < 2e-3 * abs(ref_loss), (loss.item(), ref_loss)
    assert G.rel_fro(pred.cpu(), fx[case + "pred"]) < 3e-2
    _check_bf16_samples(fx, case, m, "h14 bf16", 8e-2, 5e-2, 8e-2, sample_gate=BF16_SAMPLE_GATE_SHORT)


def _pred_sample_err(fx, case, pred, fro=False):
    flat = pred.detach().float().cpu().reshape(-1)
    stride = int(fx[case + "pred_stride"])
    return (G.rel_fro if fro else G.rel_err)(flat[::stride], fx[case + "pred_sample"]), abs(float(flat.double().norm()) / float(fx[case + "pred_norm"]) - 1.0)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("which,case,batch,seed_p,seed_x,mask_kind,train", [
    ("w12", "h14_w12/", 2, 34, 44, "random", False), ("w12", "h14_w12_train/", 2, 35, 45, "half", True), ("w32", "h14_w32/", 1, 36, 46, "random", False)])
def test_h14_grids_on_the_timed_head_dim_80_kernels_vs_reference_golden(dtype, which, case, batch, seed_p, seed_x, mask_kind, train):
    """VERDICT round 4, Missing 1: the head_dim-80 kernels `bench.py --model vit_huge` times (csrc/attn2.hip, a2::*<..., 80>; key rows of
    12..28 tokens and, on ViT-H/14's own 64 x 32 grid, exactly 32 = WP32) had kernel-level tests only -- the model-level `h14` fixtures run
    an 8 x 4 grid, which routes to the generic kernels.  Here the UNMODIFIED reference (depth 24, tests/golden/painter_h14_grids.npz:
    patch 14, embed 160 / 2 heads) on a 24 x 12 grid (eval and train mode with its recorded DropPath stream) and on the 64 x 32 grid:
    loss, pred sample, every gradient digest and sample.  The launch counters assert WHICH kernels ran: bf16 -> generation 2 for all 24
    blocks, forward and backward; fp32 (the exact build) -> the generic kernels."""
    from painter_amd import ops
    fx = G.load("painter_h14_grids.npz")
    cfg = O.h14_grid_config(which)
    m, _ = build(cfg, seed_p, dtype, train=train)
    if train:
        flat = torch.from_numpy(fx[case + "drop_scales_flat"])
        chunks = list(torch.split(flat, [int(x) for x in fx[case + "drop_scales_len"]]))
        m._drop_override = [(None, None)] + [(chunks[2 * i].cuda().contiguous(), chunks[2 * i + 1].cuda().contiguous()) for i in range(cfg.depth - 1)]
    c0 = ops.attn_launch_counts()
    loss, pred, mo, valid_d = run_painter(m, cfg, batch, seed_x, mask_kind)
    c1 = ops.attn_launch_counts()
    d = {k: tuple(b - a for a, b in zip(c0[k], c1[k])) for k in c0}
    want = (0, cfg.depth, 0) if dtype == "bf16" else (cfg.depth, 0, 0)
    assert d["fwd"] == want and d["bwd"] == want, (dtype, d)
    ref_loss = float(fx[case + "loss"])
    e_pred, e_norm = _pred_sample_err(fx, case, pred, fro=(dtype == "bf16"))
    assert mo.double().sum().item() == float(fx[case + "mask_out_sum"])
    if dtype == "fp32":
        assert abs(loss.item() - ref_loss) < 1e-4 * abs(ref_loss), (loss.item(), ref_loss)
        assert e_pred < 2e-4 and e_norm < 2e-4, (e_pred, e_norm)
        rep = []
        G.check_grad_digests(fx, case, [(n, p.grad) for n, p in m.named_parameters()], 1e-3, 1e-3, 1e-3, report=rep)
        print("h14 %s fp32: loss %.7f (reference %.7f), pred sample rel-max %.2e, worst sampled-gradient rel-max %.3e (%s)" % ((case, loss.item(), ref_loss, e_pred) + max(rep)))
    else:
        assert abs(loss.item() - ref_loss) < 2e-3 * abs(ref_loss), (loss.item(), ref_loss)
        assert e_pred < 3e-2 and e_norm < 1e-2, (e_pred, e_norm)
        print("h14 %s bf16: loss %.7f (reference %.7f), pred sample rel-fro %.2e" % (case, loss.item(), ref_loss, e_pred))
        _check_bf16_samples(fx, case, m, "h14 %s bf16 (attn2<80>)" % case, 8e-2, 5e-2, 8e-2, sample_gate=BF16_SAMPLE_GATE_SHORT)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_h14_generalised_taps_other_depth_vs_oracle(dtype):
    """depth 16 (taps 3, 7, 11, 15 = depth/4*k - 1, the extension ViT-H/14's depth 32 needs: SURVEY.md 8d note H).  No reference golden
    exists for a depth other than 24 -- the oracle with the same generalised taps is the checker; every block must receive gradient."""
    cfg = O.h14_small_config(depth=16)
    assert cfg.taps == (3, 7, 11, 15)
    m, P = build(cfg, 33, dtype)
    loss, pred, _, _ = run_painter(m, cfg, 2, 43, "random")
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    imgs, tgts, mask, valid = O.synthetic_batch(cfg, 2, 43, "random")
    lo, po, _ = O.forward(Pg, cfg, imgs, tgts, mask, valid)
    lo.backward()
    tol_l, tol_p, tol_g = (1e-4, 2e-4, 1e-3) if dtype == "fp32" else (2e-3, 3e-2, 8e-2)
    assert abs(loss.item() - lo.item()) < tol_l * abs(lo.item())
    assert G.rel_fro(pred.cpu(), po.detach()) < tol_p
    worst = max((G.rel_fro(p.grad.cpu(), Pg[n].grad), n) for n, p in m.named_parameters())
    assert worst[0] < tol_g, worst
    assert all(float(p.grad.abs().max()) > 0 for n, p in m.named_parameters() if "blocks.15." in n and "bias" not in n)


# ------------------------------------------------------------------------------------------ loss variants, pose weighting
@pytest.mark.parametrize("loss_func", ["l1", "l2", "l1l2", "smoothl1"])
def test_loss_variants_and_pose_valid_weight_vs_oracle(loss_func):
    """Painter/models_painter.py:452-460: the four loss_func branches, with a `valid` map that carries the pose task's weight 10.0
    (data/pairdataset.py:172) on a region, zeros on another and ones elsewhere: loss, pred and every parameter gradient (fp32 build)
    against the oracle's autograd."""
    import dataclasses
    cfg = dataclasses.replace(O.small_config(), loss_func=loss_func)
    m, P = build(cfg, 51, "fp32")
    imgs, tgts, mask, valid = O.synthetic_batch(cfg, 2, 61, "random")
    valid[0, :, 70:110, 8:40] = 10.0
    valid[1, :, 64:90, :] = 0.0
    for p in m.parameters():
        p.grad = None
    vd = valid.clone().cuda()
    loss, pred, _ = m(imgs.cuda(), tgts.cuda(), bool_masked_pos=mask.reshape(2, *cfg.grid).cuda(), valid=vd)
    loss.backward()
    torch.cuda.synchronize()
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    vo = valid.clone()
    lo, po, _ = O.forward(Pg, cfg, imgs, tgts, mask, vo)
    lo.backward()
    assert abs(loss.item() - lo.item()) < 1e-4 * abs(lo.item()), (loss.item(), lo.item())
    assert torch.equal(vd.cpu(), vo)                                   # the in-place ignore rule saw the same `valid`
    worst = max((G.rel_fro(p.grad.cpu(), Pg[n].grad), n) for n, p in m.named_parameters())
    assert worst[0] < 1e-3, worst
    if loss_func != "smoothl1":                                        # the weighting really matters: dropping the 10.0 changes the loss
        v1 = valid.clone().clamp_max(1.0)
        l1, _, _ = O.forward(P, cfg, imgs, tgts, mask, v1)
        assert abs(l1.item() - lo.item()) > 1e-3 * abs(lo.item())
