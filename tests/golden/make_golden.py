"""Generate the committed golden vectors by running the UNMODIFIED reference on CPU.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py            # tiny cases (seconds)
    python tests/golden/make_golden.py --full     # + ViT-L 896x448 B=1 fwd+bwd (~1-2 min, ~14 GB RSS)

The reference is imported through oracle/ref_import.py (stub modules for timm/detectron2/...,
SURVEY.md section 8c).  Parameters come from oracle.painter_oracle.random_params(cfg, seed) -- the
recipe is part of the golden contract, so fixtures store only seeds + outputs, not weights.

Outputs (tests/golden/*.npz, float32):
  loss, pred (patchified) -- full tensors for tiny cases, a strided sample + moments for ViT-L
  per-parameter gradient digests: L2 norm, sum, and dot with a seeded probe vector
  (+ full gradients of parameters with <= 4096 elements).
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
    out[prefix + "grad_names"] = np.array(names)
    out[prefix + "grad_norm"] = np.array(norms, dtype=np.float64)
    out[prefix + "grad_sum"] = np.array(sums, dtype=np.float64)
    out[prefix + "grad_dot"] = np.array(dots, dtype=np.float64)


def case_painter(cfg, out, prefix, batch, mask_kind, seed_p=1, seed_x=1234, backward=True, train_mode=False):
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true", help="also generate the ViT-L 896x448 fixture")
    ap.add_argument("--small", action="store_true", help="only (re)generate the small HIP-path fixture")
    ap.add_argument("--skip-tiny", action="store_true")
    args = ap.parse_args()
    torch.set_num_threads(os.cpu_count())

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


def _rest(args):
    if args.small or args.full:
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
