"""The reference's OWN drivers, unmodified, running on the painter_amd modules (SURVEY.md 8b "callers").

`Painter/engine_train.py:train_one_epoch` (with the reference's `util.misc.NativeScalerWithGradNormCount`, `MetricLogger`,
`util.lr_sched.adjust_learning_rate`, a plain `torch.optim.AdamW`) and `SegGPT_inference/seggpt_engine.py:run_one_image` are imported
from a reference checkout through oracle/ref_import.py (stand-ins only for the absent third-party modules) and handed OUR modules.

Needs BOTH an MI355X and the reference's driver files.  The GPU boxes of this project have no /root/reference: since round 4 the few
unmodified files the drivers consist of travel there as oracle/_ref/reference_subset.tar.gz (staged by oracle/stage_ref.py from
__graft_entry__.build() in the build container, git-ignored, SHA-256 manifest beside it) and oracle/ref_import.py unpacks them on first
use -- so these tests RUN on the GPU box.  tests/test_reference_import_cpu.py is the build-container half: the drivers import through
the stubs, every attribute of the model they touch exists on our classes, the staged copies are byte-identical to the reference."""
import types

import pytest
import torch

from oracle import painter_oracle as O
from oracle import ref_import

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not ref_import.reference_available(), reason="no reference files: neither PAINTER_REFERENCE_ROOT, /root/reference nor the staged oracle/_ref/ archive"),
              pytest.mark.skipif(not torch.cuda.is_available(), reason="needs an MI355X")]


def _small(cls, cfg, **kw):
    from functools import partial
    import torch.nn as nn
    return cls(img_size=cfg.img_size, patch_size=cfg.patch_size, embed_dim=cfg.embed_dim, depth=cfg.depth, num_heads=cfg.num_heads,
               drop_path_rate=0.1, window_size=14, qkv_bias=True, mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6),
               window_block_indexes=([0, 1], [3, 4]), residual_block_indexes=[], use_rel_pos=True, out_feature="last_feat",
               decoder_embed_dim=cfg.decoder_embed_dim, loss_func="smoothl1", **kw)


def test_reference_train_one_epoch_drives_the_hip_module():
    """engine_train.train_one_epoch (Painter/engine_train.py:34-144), unmodified: fp16 autocast context, GradScaler through the
    reference's NativeScalerWithGradNormCount, accum_iter 2, clip 3.0, lr schedule, MetricLogger -- four iterations on the small
    config; the loss must be finite, the parameters must move, and the averaged stats must come back."""
    from painter_amd import models_painter
    eng = ref_import.load_reference_engine_train()
    cfg = O.small_config()
    model = _small(models_painter.Painter, cfg, compute_dtype="bf16").cuda()
    model.load_state_dict(O.random_params(cfg, 3))
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    batches = []
    for k in range(4):
        imgs, tgts, mask, valid = O.synthetic_batch(cfg, 2, 100 + k, "random")
        batches.append((imgs, tgts, mask.reshape(2, *cfg.grid), valid))
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, betas=(0.9, 0.999), weight_decay=0.05)
    args = types.SimpleNamespace(accum_iter=2, clip_grad=3.0, lr=1e-3, min_lr=1e-5, warmup_epochs=1, epochs=2, log_wandb=False)
    stats = eng.train_one_epoch(model, batches, opt, torch.device("cuda"), 0, eng.misc.NativeScalerWithGradNormCount(),
                                log_writer=None, global_rank=0, args=args)
    assert set(stats) >= {"loss", "lr", "loss_scale", "grad_norm"} and torch.isfinite(torch.tensor(stats["loss"]))
    moved = sum(int(not torch.equal(before[n], p.detach())) for n, p in model.named_parameters())
    assert moved > 0.9 * len(before), moved


@pytest.mark.parametrize("dtype,tol_loss,tol_norm", [("fp32", 1e-3, 1e-2), ("bf16", 2e-3, 3e-2)])
def test_reference_train_one_epoch_same_losses_on_the_reference_class_and_the_hip_module(dtype, tol_loss, tol_norm):
    """VERDICT round 4 (parity soft spot d): the unmodified engine drives BOTH models -- the reference's own Painter class (PyTorch-ROCm
    eager under the engine's fp16 autocast) and ours (HIP kernels; autocast does not reach them) -- from the same parameters over the
    same four batches (accum_iter 2 -> two optimizer updates, clip 3.0, the reference's NativeScalerWithGradNormCount and lr schedule,
    plain torch.optim.AdamW).  DropPath is off in both (drop_path_rate 0: the engine puts the models in train mode and the two would
    draw different masks), everything else is the training arrangement of engine_train.py:34-144.  Compared: the four per-iteration
    losses, the two gradient norms the scaler reports at the update steps, the loss-scale trajectory and the averaged stats."""
    from painter_amd import models_painter
    eng = ref_import.load_reference_engine_train()
    refmod = ref_import.load_reference_painter()
    cfg = O.small_config()
    P = O.random_params(cfg, 3)
    batches = []
    for k in range(4):
        imgs, tgts, mask, valid = O.synthetic_batch(cfg, 2, 100 + k, "random")
        batches.append((imgs, tgts, mask.reshape(2, *cfg.grid), valid))
    args = types.SimpleNamespace(accum_iter=2, clip_grad=3.0, lr=1e-3, min_lr=1e-5, warmup_epochs=1, epochs=2, log_wandb=False)

    def drive(model):
        model.load_state_dict(P)
        losses, norms = [], []
        h = model.register_forward_hook(lambda mod, inp, out: losses.append(float(out[0].detach().float())))
        scaler = eng.misc.NativeScalerWithGradNormCount()

        class Rec:                                          # the engine calls loss_scaler(...) and loss_scaler.state_dict(): record the norms it returns
            def __call__(self, *a, **k):
                n = scaler(*a, **k)
                if n is not None:
                    norms.append(float(n))
                return n

            def state_dict(self):
                return scaler.state_dict()
        opt = torch.optim.AdamW(model.parameters(), lr=1e-3, betas=(0.9, 0.999), weight_decay=0.05)
        stats = eng.train_one_epoch(model, [tuple(t.clone() for t in b) for b in batches], opt, torch.device("cuda"), 0, Rec(), log_writer=None, global_rank=0, args=args)
        h.remove()
        return losses, norms, stats

    kw = dict(img_size=cfg.img_size, patch_size=cfg.patch_size, embed_dim=cfg.embed_dim, depth=cfg.depth, num_heads=cfg.num_heads,
              drop_path_rate=0.0, window_size=14, qkv_bias=True, mlp_ratio=4, window_block_indexes=([0, 1], [3, 4]), residual_block_indexes=[],
              use_rel_pos=True, out_feature="last_feat", decoder_embed_dim=cfg.decoder_embed_dim, loss_func="smoothl1")
    from functools import partial
    import torch.nn as nn
    kw["norm_layer"] = partial(nn.LayerNorm, eps=1e-6)
    lr_, nr_, sr_ = drive(refmod.Painter(**kw).cuda())
    lo_, no_, so_ = drive(models_painter.Painter(compute_dtype=dtype, **kw).cuda())
    print("train_one_epoch, reference class vs HIP module (%s): losses %s vs %s; grad norms %s vs %s; loss scale %s vs %s"
          % (dtype, ["%.6f" % v for v in lr_], ["%.6f" % v for v in lo_], ["%.4f" % v for v in nr_], ["%.4f" % v for v in no_], sr_["loss_scale"], so_["loss_scale"]))
    assert len(lr_) == len(lo_) == 4 and len(nr_) == len(no_) == 2
    for a, b in zip(lr_, lo_):
        assert abs(a - b) <= tol_loss * abs(a), (lr_, lo_)
    for a, b in zip(nr_, no_):
        assert abs(a - b) <= tol_norm * abs(a), (nr_, no_)
    assert sr_["loss_scale"] == so_["loss_scale"] and abs(sr_["lr"] - so_["lr"]) < 1e-12
    assert abs(sr_["loss"] - so_["loss"]) <= tol_loss * abs(sr_["loss"]) and abs(sr_["grad_norm"] - so_["grad_norm"]) <= tol_norm * abs(sr_["grad_norm"])


def test_reference_run_one_image_drives_the_hip_seggpt_module():
    """seggpt_engine.run_one_image (SegGPT_inference/seggpt_engine.py:26-53), unmodified, on our SegGPT module: two prompts over one
    query (feature ensemble on), float64 host arrays in, a de-normalised [H/2, W, 3] picture out; checked against the CPU oracle."""
    import numpy as np
    from painter_amd import models_seggpt
    eng = ref_import.load_reference_seggpt_engine()
    cfg = O.small_config(seggpt=True)
    model = _small(models_seggpt.SegGPT, cfg, compute_dtype="fp32").cuda().eval()
    P = O.random_params(cfg, 5)
    model.load_state_dict(P)
    model.seg_type = "instance"
    imgs, tgts, _, _ = O.synthetic_batch(cfg, 2, 9, "half")
    img = imgs.permute(0, 2, 3, 1).double().numpy()
    tgt = tgts.permute(0, 2, 3, 1).double().numpy()
    out = eng.run_one_image(img, tgt, model, torch.device("cuda"))
    L = cfg.grid[0] * cfg.grid[1]
    mask = torch.zeros(1, L)
    mask[:, L // 2:] = 1
    with torch.no_grad():
        _, yo, _ = O.forward(P, cfg, imgs, tgts, mask, torch.ones_like(tgts), torch.ones(2, 1), 0)
    y = O.unpatchify(yo, cfg.patch_size).permute(0, 2, 3, 1)
    ref = torch.clip((y[0, y.shape[1] // 2:] * torch.tensor(O.IMAGENET_STD) + torch.tensor(O.IMAGENET_MEAN)) * 255, 0, 255)
    assert out.shape == ref.shape and float((out - ref).abs().max()) < 0.05


def test_bench_reference_gpu_leg_times_the_unmodified_model_and_agrees_with_the_hip_path():
    """bench.py's `reference_gpu` leg (the unmodified reference ViT-L through PyTorch-ROCm eager on this GPU, the bench model's parameters
    and batch): it must run (bf16 and fp16 autocast), and -- in eval mode, where DropPath draws nothing -- the loss it reports under
    bf16 autocast must be the loss of the measured path on the same batch (both bf16: 2e-3, the gate of the bf16 parity tests)."""
    import bench
    from painter_amd import models_painter
    dev = torch.device("cuda", 0)
    model = models_painter.painter_vit_large_patch16_input896x448_win_dec64_8glb_sl1(compute_dtype="bf16")
    bench.randomize_parameters(model, seed=1)
    model = model.to(dev).eval()
    c = model._cfg
    inputs = bench.synthetic_inputs(1, c.H, c.W, c.L, 77, dev)
    with torch.no_grad():
        ours = float(model(inputs[0], inputs[1], bool_masked_pos=inputs[2], valid=inputs[3].clone())[0])
    rg = bench.reference_gpu_baseline(model, inputs, dev, steps=1, warmup=1, train=False)
    assert rg is not None and rg["kind"] == "reference" and rg["value"] > 0 and rg["fp16_autocast_value"] > 0
    print("reference on this GPU, B=1 eval-mode parameters: %.1f images/s (bf16 autocast); loss %.6f vs the HIP path %.6f"
          % (rg["value"], rg["loss_bf16"], ours))
    assert abs(rg["loss_bf16"] - ours) <= 2e-3 * abs(ours), (rg["loss_bf16"], ours)
