"""bf16 gates measured against the reference ITSELF, live, in the test process (parity soft spots 1b and 1c).

`tests/test_model_gpu.py` holds the bf16 build to the reference's own bf16-autocast deviation at ViT-L (a number measured once by
tools/grad_yardstick.py and written into the gates).  Two shapes had no such yardstick:

* the head_dim-80 models (patch 14, embed 160 / 2 heads, depth 24): an absolute 1e-1 on the sampled gradients, 2.4 x above what the build
  measures.  Here the UNMODIFIED reference (oracle/ref_import.py; on the GPU box the staged subset of oracle/stage_ref.py) runs the same
  case in fp32 and under torch.autocast(bfloat16) -- the arrangement of engine_train.py:65-75 -- on the GPU, and the HIP bf16 build must not
  deviate more from the fp32 run than the reference's own bf16 run does, metric by metric;
* ViT-L under the reference's own engine: `train_one_epoch` (engine_train.py:34-144), unmodified, driving the reference class and the HIP
  module side by side had been run on the small configuration only.

Needs an MI355X and the reference's files (staged archive or checkout)."""
import types
from functools import partial

import pytest
import torch
import torch.nn as nn

from oracle import painter_oracle as O
from oracle import ref_import
from tests import golden_util as G

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not ref_import.reference_available(), reason="no reference files: neither PAINTER_REFERENCE_ROOT, /root/reference nor the staged oracle/_ref/ archive"),
              pytest.mark.skipif(not torch.cuda.is_available(), reason="needs an MI355X")]

STRIDE, SMALL = 997, 4096          # the sampling of tests/golden/make_golden.py


def _kwargs(cfg, drop_path_rate=0.1):
    return dict(img_size=cfg.img_size, patch_size=cfg.patch_size, embed_dim=cfg.embed_dim, depth=cfg.depth, num_heads=cfg.num_heads,
                drop_path_rate=drop_path_rate, window_size=14, qkv_bias=True, mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6),
                window_block_indexes=([0, 1], [3, 4]), residual_block_indexes=[], use_rel_pos=True, out_feature="last_feat",
                decoder_embed_dim=cfg.decoder_embed_dim, loss_func="smoothl1")


def _grads(model, args, autocast=None):
    model.zero_grad(set_to_none=True)
    if autocast is None:
        loss, pred, _ = model(*args)
    else:
        with torch.autocast("cuda", dtype=autocast):
            loss, pred, _ = model(*args)
    loss.backward()
    torch.cuda.synchronize()
    return float(loss.detach().float()), pred.detach().float().cpu(), {n: p.grad.detach().float().cpu().clone() for n, p in model.named_parameters()}


def _deviation(g, base):
    """The metrics of tests/test_model_gpu.py::_check_bf16_samples: worst sampled rel-max (rel-pos tables apart) and worst whole-tensor
    relative Frobenius error (rel-pos tables; every other tensor with more than one dimension)."""
    out = {"sample_relpos": (0.0, "-"), "sample_other": (0.0, "-"), "fro_relpos": (0.0, "-"), "fro_matrices": (0.0, "-")}
    for n, b in base.items():
        a, b = g[n].reshape(-1), b.reshape(-1)
        is_rel = n.endswith("rel_pos_h") or n.endswith("rel_pos_w")
        e = G.rel_err(a[::STRIDE], b[::STRIDE]) if b.numel() > SMALL else G.rel_err(a, b)
        k = "sample_relpos" if is_rel else "sample_other"
        out[k] = max(out[k], (e, n))
        if is_rel or base[n].dim() > 1:
            k = "fro_relpos" if is_rel else "fro_matrices"
            out[k] = max(out[k], (G.rel_fro(a, b), n))
    return out


# HIP bf16 deviation <= FACTOR x the reference's own bf16-autocast deviation from the same fp32 gradients.  Measured (round 6,
# profiles/r06_live_yardstick_head_dim_80_and_vit_large_engine.log): the HIP build sits at 0.55 - 0.80 x the reference's own deviation on every
# metric of the three cases (worst: sampled rel-max on the 64 x 32 grid, 2.40e-2 against 3.02e-2) -- it rounds to bf16 once per GEMM where
# autocast rounds after every op -- so the gate is 1.0 x: never worse than the reference itself.
FACTOR_SAMPLE, FACTOR_FRO = 1.0, 1.0


@pytest.mark.parametrize("which,batch,seed_p,seed_x", [("small", 2, 31, 41), ("w12", 2, 34, 44), ("w32", 1, 36, 46)])
def test_head_dim_80_bf16_build_within_the_reference_own_bf16_deviation(which, batch, seed_p, seed_x):
    """head_dim 80 / patch 14 at depth 24 (the depth the unmodified class can run): the 8 x 4 grid of painter_h14.npz (generic kernels), the
    24 x 12 and 64 x 32 grids of painter_h14_grids.npz (the head_dim-80 kernels `bench.py --model vit_huge` times)."""
    from painter_amd import models_painter
    cfg = O.h14_small_config(depth=24) if which == "small" else O.h14_grid_config(which)
    P = O.random_params(cfg, seed_p)
    imgs, tgts, mask, valid = O.synthetic_batch(cfg, batch, seed_x, "random")
    args = (imgs.cuda(), tgts.cuda(), mask.reshape(batch, *cfg.grid).cuda(), valid.cuda())
    rm = ref_import.load_reference_painter().Painter(**_kwargs(cfg))
    rm.load_state_dict(P, strict=True)
    rm = rm.cuda().eval()
    l32, p32, g32 = _grads(rm, tuple(t.clone() for t in args))
    l16, p16, g16 = _grads(rm, tuple(t.clone() for t in args), torch.bfloat16)
    del rm
    m = models_painter.Painter(compute_dtype="bf16", **_kwargs(cfg))
    m.load_state_dict(P, strict=True)
    m = m.cuda().eval()
    m.zero_grad(set_to_none=True)
    loss, pred, _ = m(args[0], args[1], bool_masked_pos=args[2], valid=args[3].clone())
    loss.backward()
    torch.cuda.synchronize()
    gh = {n: p.grad.detach().float().cpu() for n, p in m.named_parameters()}
    ref, hip = _deviation(g16, g32), _deviation(gh, g32)
    pr, ph = G.rel_fro(p16, p32), G.rel_fro(pred.detach().float().cpu(), p32)
    lr, lh = abs(l16 - l32) / abs(l32), abs(float(loss.detach()) - l32) / abs(l32)
    print("head_dim 80 (%s): reference bf16 autocast | HIP bf16, both against the reference's fp32 run: loss %.2e | %.2e, pred rel-Frobenius %.2e | %.2e; "
          % (which, lr, lh, pr, ph) + "; ".join("%s %.3e (%s) | %.3e (%s)" % ((k,) + ref[k] + hip[k]) for k in ref))
    assert ph <= FACTOR_FRO * pr, (ph, pr)
    assert lh <= max(2.0 * lr, 2e-3), (lh, lr)
    for k in ref:
        f = FACTOR_SAMPLE if k.startswith("sample") else FACTOR_FRO
        assert hip[k][0] <= f * ref[k][0], (k, hip[k], ref[k])


def test_reference_train_one_epoch_at_vit_large_same_losses_on_the_reference_class_and_the_hip_module():
    """tests/test_reference_engine_gpu.py's side-by-side run at the headline configuration's sizes: ViT-L 896 x 448, B = 1, four iterations
    (accum_iter 2 -> two optimizer updates, clip 3.0, the reference's NativeScalerWithGradNormCount and lr schedule, torch.optim.AdamW), the
    unmodified engine driving the unmodified class (fp16 autocast, as engine_train.py:65) and the HIP bf16 module from the same parameters
    over the same batches; DropPath off in both (the two would draw different masks).  The second pair of iterations runs on parameters each
    model updated ITSELF: the losses agreeing there is the optimizer-in-the-loop statement the fixtures cannot make."""
    from painter_amd import models_painter
    eng = ref_import.load_reference_engine_train()
    refmod = ref_import.load_reference_painter()
    cfg = O.vit_large_config()
    P = O.random_params(cfg, 3)
    batches = []
    for k in range(4):
        imgs, tgts, mask, valid = O.synthetic_batch(cfg, 1, 100 + k, "random")
        batches.append((imgs, tgts, mask.reshape(1, *cfg.grid), valid))
    args = types.SimpleNamespace(accum_iter=2, clip_grad=3.0, lr=1e-4, min_lr=1e-6, warmup_epochs=1, epochs=2, log_wandb=False)

    def drive(model):
        model.load_state_dict(P)
        losses, norms = [], []
        h = model.register_forward_hook(lambda mod, inp, out: losses.append(float(out[0].detach().float())))
        scaler = eng.misc.NativeScalerWithGradNormCount()

        class Rec:
            def __call__(self, *a, **k):
                n = scaler(*a, **k)
                if n is not None:
                    norms.append(float(n))
                return n

            def state_dict(self):
                return scaler.state_dict()
        opt = torch.optim.AdamW(model.parameters(), lr=1e-4, betas=(0.9, 0.999), weight_decay=0.05)
        eng.train_one_epoch(model, [tuple(t.clone() for t in b) for b in batches], opt, torch.device("cuda"), 0, Rec(), log_writer=None, global_rank=0, args=args)
        h.remove()
        moved = model.state_dict()["blocks.0.attn.qkv.weight"].detach().float().cpu().clone()
        return losses, norms, moved

    lr_, nr_, wr_ = drive(refmod.Painter(**_kwargs(cfg, 0.0)).cuda())
    torch.cuda.empty_cache()
    lo_, no_, wo_ = drive(models_painter.Painter(compute_dtype="bf16", **_kwargs(cfg, 0.0)).cuda())
    step = G.rel_fro(wo_ - P["blocks.0.attn.qkv.weight"], wr_ - P["blocks.0.attn.qkv.weight"])
    print("train_one_epoch at ViT-L, reference class (fp16 autocast) vs HIP bf16 module: losses %s vs %s; grad norms %s vs %s; relative difference of "
          "the two updates of blocks.0.attn.qkv.weight %.3e" % (["%.6f" % v for v in lr_], ["%.6f" % v for v in lo_], ["%.4f" % v for v in nr_], ["%.4f" % v for v in no_], step))
    assert len(lr_) == len(lo_) == 4 and len(nr_) == len(no_) == 2
    for a, b in zip(lr_, lo_):
        assert abs(a - b) <= 2e-4 * abs(a), (lr_, lo_)             # measured 9e-6
    for a, b in zip(nr_, no_):
        assert abs(a - b) <= 3e-3 * abs(a), (nr_, no_)             # measured 3e-4
