"""Generate the committed golden vectors by running the UNMODIFIED reference on CPU.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py            # tiny cases (seconds)
    python tests/golden/make_golden.py --full     # + ViT-L 896x448 B=1 fwd+bwd (~1-2 min, ~14 GB RSS)

The reference is imported through oracle/ref_import.py (stub modules for timm/detectron2/...,
SURVEY.md section 8c).  Parameters come from oracle.painter_oracle.random_params(cfg, seed) -- the
recipe is part of the golden contract, so fixtures store only seeds + outputs, not weights.

Outputs (tests/golden/*.npz, float32):
  loss, pred (patchified) -- full tensors for tiny cases, a strided sample + moments for ViT-L
  per-parameter gradient digests: L2 norm, sum, and dot with a seeded probe vector; full gradients of parameters with
  <= 4096 elements; every 997th element of every larger gradient (grad_sample/<name>).
"""
import argparse
import os
import sys
from functools import partial

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import painter_oracle as O      # noqa: E402
from oracle import ref_import               # noqa: E402

WINDOW_BLOCK_INDEXES = (list(range(0, 2)) + list(range(3, 5)) + list(range(6, 8)) + list(range(9, 11)) +
                        list(range(12, 14)), list(range(15, 17)), list(range(18, 20)), list(range(21, 23)))
PRED_STRIDE = 37          # ViT-L pred sample stride (flattened patchified pred)
SMALL_PARAM = 4096
GRAD_STRIDE = 997         # every 997th element (a prime: no aliasing with any row width) of every gradient with > SMALL_PARAM elements


def build_reference(cfg: O.OracleConfig, seed: int):
    """Instantiate the reference class exactly as its factory does (models_painter.py:476-487 /
    models_seggpt.py:483-494) but with cfg's sizes, then load the seeded parameters."""
    if cfg.seggpt:
        mod = ref_import.load_reference_seggpt()
        cls = mod.SegGPT
    else:
        mod = ref_import.load_reference_painter()
        cls = mod.Painter
    kw = dict(img_size=cfg.img_size, patch_size=cfg.patch_size, embed_dim=cfg.embed_dim, depth=cfg.depth,
              num_heads=cfg.num_heads, drop_path_rate=0.1, window_size=14, qkv_bias=True, mlp_ratio=4,
              norm_layer=partial(nn.LayerNorm, eps=1e-6), window_block_indexes=WINDOW_BLOCK_INDEXES,
              residual_block_indexes=[], use_rel_pos=True, out_feature="last_feat",
              decoder_embed_dim=cfg.decoder_embed_dim, loss_func="smoothl1")
    model = cls(**kw)
    P = O.random_params(cfg, seed)
    sd = model.state_dict()
    assert list(sd.keys()) == list(P.keys()), "state_dict ABI drifted from oracle.param_shapes"
    for k in sd:
        assert tuple(sd[k].shape) == tuple(P[k].shape), k
    model.load_state_dict(P, strict=True)
    return model, P


def probe_vector(name: str, numel: int) -> torch.Tensor:
    """Deterministic probe for the grad digests (seed from the parameter name)."""
    s = sum((i + 1) * ord(c) for i, c in enumerate(name)) % (2 ** 31)
    g = torch.Generator().manual_seed(s)
    return torch.randn(numel, generator=g, dtype=torch.float32)


def grad_digest(named_grads, out: dict, prefix: str):
    names, norms, sums, dots = [], [], [], []
    for name, g in named_grads:
        g = g.detach().float().reshape(-1)
        names.append(name)
        norms.append(float(g.double().norm()))
        sums.append(float(g.double().sum()))
        dots.append(float((g.double() * probe_vector(name, g.numel()).double()).sum()))
        if g.numel() <= SMALL_PARAM:
            out[f"{prefix}grad/{name}"] = g.numpy()
        else:
            # a strided sample of the tensor itself: unlike the norm / probe-dot digests (a random direction of the right norm
            # changes the probe dot by only ~1.4/sqrt(n) of the scale) this can tell a wrong gradient from a right one
            out[f"{prefix}grad_sample/{name}"] = g[::GRAD_STRIDE].clone().numpy()
            # the rel-pos tables are short ([111, 64] / [55, 64]: 8 / 4 samples at stride 997) and the noisiest gradients of the bf16 build:
            # they are stored whole as well, so that the test can take a relative Frobenius error over the full tensor (round 4)
            if name.endswith("rel_pos_h") or name.endswith("rel_pos_w"):
                out[f"{prefix}grad_full/{name}"] = g.clone().numpy()
    out[prefix + "grad_names"] = np.array(names)
    out[prefix + "grad_norm"] = np.array(norms, dtype=np.float64)
    out[prefix + "grad_sum"] = np.array(sums, dtype=np.float64)
    out[prefix + "grad_dot"] = np.array(dots, dtype=np.float64)
    out[prefix + "grad_sample_stride"] = np.int64(GRAD_STRIDE)


def case_painter(cfg, out, prefix, batch, mask_kind, seed_p=1, seed_x=1234, backward=True, train_mode=False, pred_stride=0):
    model, _ = build_reference(cfg, seed_p)
    imgs, tgts, mask, valid = O.synthetic_batch(cfg, batch, seed_x, mask_kind)
    if train_mode:
        model.train()
        ref_import._DropPath.record = rec = []
        orig = ref_import._DropPath.forward

        def fwd(self, x):
            if self.drop_prob == 0.0 or not self.training:
                return x
            keep = 1 - self.drop_prob
            shape = (x.shape[0],) + (1,) * (x.ndim - 1)
            rt = (keep + torch.rand(shape, dtype=x.dtype, device=x.device)).floor_()
            rec.append((rt.reshape(-1) / keep).clone())
            return x.div(keep) * rt
        ref_import._DropPath.forward = fwd
        torch.manual_seed(7)
    else:
        model.eval()
    for p in model.parameters():
        p.grad = None
    loss, pred, m = model(imgs, tgts, bool_masked_pos=mask.reshape(batch, *cfg.grid), valid=valid)
    out[prefix + "loss"] = np.float64(loss.item())
    if pred_stride:         # big grids: a strided sample + the norm instead of the tensor (as the ViT-L fixtures do)
        flat = pred.detach().reshape(-1)
        out[prefix + "pred_sample"] = flat[::pred_stride].numpy()
        out[prefix + "pred_stride"] = np.int64(pred_stride)
        out[prefix + "pred_norm"] = np.float64(flat.double().norm().item())
        out[prefix + "mask_out_sum"] = np.float64(m.double().sum().item())
    else:
        out[prefix + "pred"] = pred.detach().numpy()
        out[prefix + "mask_out"] = m.numpy()
    out[prefix + "valid_out_sum"] = np.float64(valid.double().sum().item())
    if backward:
        loss.backward()
        grad_digest([(n, p.grad) for n, p in model.named_parameters()], out, prefix)
    if train_mode:
        ref_import._DropPath.forward = orig
        # block 0 has drop_path == 0 -> nn.Identity (models_painter.py:199): two calls per block otherwise
        # ragged in B' (2B for idx<=2, B after) -> store flat + lengths
        out[prefix + "drop_scales_flat"] = torch.cat([r for r in rec]).numpy()
        out[prefix + "drop_scales_len"] = np.array([r.numel() for r in rec])


def case_seggpt(cfg, out, prefix, n_prompts, merge_between_batch, seg_kind, seed_p=2, seed_x=4321):
    model, _ = build_reference(cfg, seed_p)
    model.eval()
    imgs, tgts, _, valid = O.synthetic_batch(cfg, n_prompts, seed_x, "half")
    L = cfg.grid[0] * cfg.grid[1]
    mask = torch.zeros(L)
    mask[L // 2:] = 1
    mask = mask.unsqueeze(0)                 # [1, L] broadcast over prompts (seggpt_engine.py:36-38)
    seg_type = torch.ones(n_prompts, 1) if seg_kind == "instance" else torch.zeros(n_prompts, 1)
    with torch.no_grad():
        loss, pred, m = model(imgs, tgts, mask, valid, seg_type, merge_between_batch)
    out[prefix + "loss"] = np.float64(loss.item())
    out[prefix + "pred"] = pred.numpy()


def case_vit_large(out, prefix):
    cfg = O.vit_large_config()
    model, _ = build_reference(cfg, 1)
    model.eval()
    imgs, tgts, mask, valid = O.synthetic_batch(cfg, 1, 1234, "random")
    loss, pred, m = model(imgs, tgts, bool_masked_pos=mask.reshape(1, *cfg.grid), valid=valid)
    out[prefix + "loss"] = np.float64(loss.item())
    flat = pred.detach().reshape(-1)
    out[prefix + "pred_sample"] = flat[::PRED_STRIDE].numpy()
    out[prefix + "pred_stride"] = np.int64(PRED_STRIDE)
    out[prefix + "pred_mean"] = np.float64(flat.double().mean().item())
    out[prefix + "pred_norm"] = np.float64(flat.double().norm().item())
    loss.backward()
    grad_digest([(n, p.grad) for n, p in model.named_parameters()], out, prefix)


def _record_droppath(rec):
    """Replace the DropPath stand-in's forward by one that records the per-sample factors it draws (timm 0.3.2 semantics)."""
    orig = ref_import._DropPath.forward

    def fwd(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1 - self.drop_prob
        shape = (x.shape[0],) + (1,) * (x.ndim - 1)
        rt = (keep + torch.rand(shape, dtype=x.dtype, device=x.device)).floor_()
        rec.append((rt.reshape(-1) / keep).clone())
        return x.div(keep) * rt
    ref_import._DropPath.forward = fwd
    return orig


def case_vit_large_b8_train(out, prefix, batch=8):
    """BASELINE configs[1]: ViT-L 896x448, B = 8, TRAIN mode (DropPath 0.1), forward + backward.

    The unmodified reference at B = 8 needs > 100 GB (it keeps every fp32 attention matrix), the build container has 62 GB.  Samples
    do not interact in Painter.forward except through the loss normaliser (models_painter.py:462: sum(loss * mask) / (sum(mask) + 1e-2)
    over the whole batch) and DropPath draws one factor per sample and residual branch, so the B = 8 result is assembled from eight
    B = 1 runs of the unmodified reference:  S_b = loss_b * (M_b + 1e-2),  loss = sum_b S_b / (sum_b M_b + 1e-2),
    grad = sum_b grad_b * (M_b + 1e-2) / (sum_b M_b + 1e-2); pred rows are the runs' own.  The recorded DropPath factors are stored in
    the order a B = 8 forward consumes them (blocks 0..2 run on the concatenated [x ; y] batch: x-stream factors of all samples,
    then y-stream factors).  tests/test_model_gpu.py checks the fp32 build against this at 1e-3 (which also pins the assembly
    arithmetic) and the bf16 build at its stated tolerances."""
    cfg = O.vit_large_config()
    model, _ = build_reference(cfg, 1)
    model.train()
    imgs, tgts, mask, valid = O.synthetic_batch(cfg, batch, 4321, "random")
    names = [n for n, _ in model.named_parameters()]
    S, M, recs, preds = [], [], [], []
    gsum = None
    torch.manual_seed(11)
    for b_ in range(batch):
        rec = []
        orig = _record_droppath(rec)
        for p in model.parameters():
            p.grad = None
        v = valid[b_:b_ + 1].clone()
        loss, pred, _ = model(imgs[b_:b_ + 1], tgts[b_:b_ + 1], bool_masked_pos=mask[b_:b_ + 1].reshape(1, *cfg.grid), valid=v)
        loss.backward()
        ref_import._DropPath.forward = orig
        # sum(mask) of this sample: mask (0/1 per patch) x patch pixels x 3 channels x valid (ones here, no sample is ignored)
        Mb = float(mask[b_].double().sum()) * cfg.patch_size ** 2 * 3
        assert float(v.double().min()) == 1.0, "ignore rule fired: the assembly below assumes it does not"
        S.append(loss.item() * (Mb + 1e-2))
        M.append(Mb)
        recs.append(rec)
        preds.append(pred.detach().reshape(-1)[::PRED_STRIDE].clone())
        g = [p.grad.detach().double() * (Mb + 1e-2) for p in model.parameters()]
        gsum = g if gsum is None else [a + c for a, c in zip(gsum, g)]
        print("  sample", b_, "loss", loss.item(), flush=True)
    den = sum(M) + 1e-2
    out[prefix + "loss"] = np.float64(sum(S) / den)
    out[prefix + "pred_sample"] = torch.stack(preds).numpy()
    out[prefix + "pred_stride"] = np.int64(PRED_STRIDE)
    grad_digest([(n, (g / den).float()) for n, g in zip(names, gsum)], out, prefix)
    # DropPath factors in B = 8 consumption order: per recorded call k (2 per block for blocks 1..23), concat over samples;
    # calls with 2 factors per sample (blocks <= 2, the [x ; y] batch) become [x of all samples, y of all samples]
    ncall = len(recs[0])
    flat, lens = [], []
    for k in range(ncall):
        per = [recs[b_][k] for b_ in range(batch)]
        if per[0].numel() == 2:
            v = torch.cat([torch.stack([p_[0] for p_ in per]), torch.stack([p_[1] for p_ in per])])
        else:
            v = torch.cat(per)
        flat.append(v)
        lens.append(v.numel())
    out[prefix + "drop_scales_flat"] = torch.cat(flat).numpy()
    out[prefix + "drop_scales_len"] = np.array(lens)


def case_seggpt_vit_large_n32(out, prefix, n_prompts=32):
    """BASELINE configs[3]: seggpt_vit_large_patch16_input896x448, 32 in-context prompts sharing one query, feature ensemble from block 0
    (merge_between_batch = 0), seg_type ones, bottom-half mask -- forward of the unmodified reference (models_seggpt.py:471-479)."""
    cfg = O.vit_large_config(seggpt=True)
    model, _ = build_reference(cfg, 2)
    model.eval()
    imgs, tgts, _, valid = O.synthetic_batch(cfg, n_prompts, 777, "half")
    # one query under all prompts: the query half (rows 448..895) of imgs is the same picture for every prompt (seggpt_engine.py:75-90)
    imgs[:, :, cfg.img_size[0] // 2:, :] = imgs[0:1, :, cfg.img_size[0] // 2:, :]
    L = cfg.grid[0] * cfg.grid[1]
    mask = torch.zeros(1, L)
    mask[:, L // 2:] = 1
    seg_type = torch.ones(n_prompts, 1)
    with torch.no_grad():
        loss, pred, _ = model(imgs, tgts, mask, valid, seg_type, 0)
    out[prefix + "loss"] = np.float64(loss.item())
    flat = pred.reshape(n_prompts, -1)
    out[prefix + "pred_sample"] = flat[:, ::PRED_STRIDE].numpy()
    out[prefix + "pred_stride"] = np.int64(PRED_STRIDE)
    out[prefix + "pred_norm"] = flat.double().norm(dim=1).numpy()


def case_h14(out):
    """hd = 80 / patch 14 arithmetic pinned to the UNMODIFIED reference: Painter(img_size=(112, 56), patch_size=14, embed_dim=160,
    depth=24, num_heads=2) -- head_dim 80, an 8 x 4 token grid, the 16 x 16 pre-training position grid of patch 14 -- at depth 24, where
    the reference's hard-coded taps [5, 11, 17, 23] (models_painter.py:416) ARE the generalised depth/4*k - 1.  (ViT-H/14 proper, depth
    32, has no reference golden: its taps are an extension, SURVEY.md 8d note H.)"""
    cfg = O.h14_small_config(depth=24)
    case_painter(cfg, out, "h14_rand/", batch=2, mask_kind="random", seed_p=31, seed_x=41)
    case_painter(cfg, out, "h14_train/", batch=2, mask_kind="half", seed_p=32, seed_x=42, train_mode=True)


def case_h14_grids(out):
    """The head_dim-80 / patch-14 arithmetic on the token grids whose attention runs on the kernels `bench.py --model vit_huge` times
    (csrc/attn2.hip<..., 80>: key rows of 12..28 tokens, and exactly 32 = the WP32 path of ViT-H/14's own 64 x 32 grid) -- the 8 x 4
    grid of case_h14 routes to the generic kernels.  UNMODIFIED reference, depth 24 (its hard-coded taps), eval and train mode."""
    w12, w32 = O.h14_grid_config("w12"), O.h14_grid_config("w32")
    case_painter(w12, out, "h14_w12/", batch=2, mask_kind="random", seed_p=34, seed_x=44, pred_stride=7)
    case_painter(w12, out, "h14_w12_train/", batch=2, mask_kind="half", seed_p=35, seed_x=45, train_mode=True, pred_stride=7)
    case_painter(w32, out, "h14_w32/", batch=1, mask_kind="random", seed_p=36, seed_x=46, pred_stride=37)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true", help="also generate the ViT-L 896x448 fixture")
    ap.add_argument("--small", action="store_true", help="only (re)generate the small HIP-path fixture")
    ap.add_argument("--skip-tiny", action="store_true")
    ap.add_argument("--vitl-b8", action="store_true", help="only: ViT-L B=8 train-mode fixture (8 B=1 runs of the reference, ~3 min, 15 GB)")
    ap.add_argument("--seggpt-n32", action="store_true", help="only: SegGPT ViT-L N=32 feature-ensemble forward (~6 min)")
    ap.add_argument("--h14", action="store_true", help="only: the head_dim 80 / patch 14 small fixture (seconds)")
    ap.add_argument("--h14-grids", action="store_true", help="only: head_dim 80 / patch 14 on the 24 x 12 and 64 x 32 token grids (the timed attn2<80> kernels)")
    ap.add_argument("--vitl-b1", action="store_true", help="only: ViT-L B=1 eval fixture (~2 min)")
    args = ap.parse_args()
    torch.set_num_threads(os.cpu_count())
    if args.vitl_b8:
        out = {}
        case_vit_large_b8_train(out, "vitl_b8_train/")
        np.savez_compressed(os.path.join(HERE, "painter_vitl_b8.npz"), **out)
        print("painter_vitl_b8.npz loss", out["vitl_b8_train/loss"])
        return
    if args.h14:
        out = {}
        case_h14(out)
        np.savez_compressed(os.path.join(HERE, "painter_h14.npz"), **out)
        print("painter_h14.npz", {k: v for k, v in out.items() if k.endswith("loss")})
        return
    if args.h14_grids:
        out = {}
        case_h14_grids(out)
        np.savez_compressed(os.path.join(HERE, "painter_h14_grids.npz"), **out)
        print("painter_h14_grids.npz", {k: v for k, v in out.items() if k.endswith("loss")})
        return
    if args.vitl_b1:
        args.full = True
        return _rest(args, only_vitl=True)
    if args.seggpt_n32:
        out = {}
        case_seggpt_vit_large_n32(out, "seggpt_n32/")
        np.savez_compressed(os.path.join(HERE, "seggpt_vitl_n32.npz"), **out)
        print("seggpt_vitl_n32.npz loss", out["seggpt_n32/loss"])
        return

    if args.skip_tiny or args.small:
        return _rest(args)
    out = {}
    tiny = O.tiny_config()
    case_painter(tiny, out, "painter_half/", batch=2, mask_kind="half")
    case_painter(tiny, out, "painter_rand/", batch=3, mask_kind="random", seed_p=3, seed_x=99)
    case_painter(tiny, out, "painter_train/", batch=2, mask_kind="random", seed_p=5, seed_x=11, train_mode=True)
    np.savez_compressed(os.path.join(HERE, "painter_tiny.npz"), **out)
    print("painter_tiny.npz", {k: v for k, v in out.items() if k.endswith("loss")})

    out = {}
    tseg = O.tiny_config(seggpt=True)
    case_seggpt(tseg, out, "seggpt_n1/", 1, -1, "semantic")
    case_seggpt(tseg, out, "seggpt_n3_merge/", 3, 0, "instance")
    case_seggpt(tseg, out, "seggpt_n4_merge/", 4, 0, "semantic", seed_p=4, seed_x=77)
    np.savez_compressed(os.path.join(HERE, "seggpt_tiny.npz"), **out)
    print("seggpt_tiny.npz", {k: v for k, v in out.items() if k.endswith("loss")})

    _rest(args)


def _rest(args, only_vitl=False):
    if (args.small or args.full) and not only_vitl:
        out = {}
        small = O.small_config()
        case_painter(small, out, "painter_rand/", batch=2, mask_kind="random", seed_p=11, seed_x=21)
        case_painter(small, out, "painter_train/", batch=2, mask_kind="half", seed_p=12, seed_x=22, train_mode=True)
        ssmall = O.small_config(seggpt=True)
        case_seggpt(ssmall, out, "seggpt_n3_merge/", 3, 0, "instance", seed_p=13, seed_x=23)
        case_seggpt(ssmall, out, "seggpt_n1/", 1, -1, "semantic", seed_p=13, seed_x=24)
        np.savez_compressed(os.path.join(HERE, "painter_small.npz"), **out)
        print("painter_small.npz", {k: v for k, v in out.items() if k.endswith("loss")})

    if args.full:
        out = {}
        case_vit_large(out, "vitl_b1/")
        np.savez_compressed(os.path.join(HERE, "painter_vitl.npz"), **out)
        print("painter_vitl.npz loss", out["vitl_b1/loss"])


if __name__ == "__main__":
    main()
