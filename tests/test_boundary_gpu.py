"""The drop-in boundary driven the way the reference's own training loop drives it.

/root/reference does not exist on the GPU box, so the loop below RE-CREATES the call pattern of Painter/engine_train.py:51-103 and
Painter/util/misc.py:252-278 (it is not a copy of those files): `model.train(True)`, `optimizer.zero_grad()`, per iteration the
forward inside `torch.cuda.amp.autocast()` with an int32 [B, Hp, Wp] `bool_masked_pos` and a float `valid`, `loss.item()` + the
finiteness check, `loss /= accum_iter`, torch's own GradScaler (`scale(loss).backward()`, on update steps `unscale_` ->
`clip_grad_norm_(model.parameters(), 3.0)` -> `scaler.step(optimizer)` -> `scaler.update()`), `optimizer.zero_grad()` on update
steps, `torch.cuda.synchronize()`; the optimizer is a plain torch.optim.AdamW over the module's parameters, as main_train.py:344-348
builds it.  Run once on the bare module and once under `DistributedDataParallel(model)` (find_unused_parameters=False,
main_train.py:340) on a 1-rank RCCL group; both must track the same loop executed on the CPU oracle."""
import math
import os
import socket
from functools import partial

import pytest
import torch
import torch.nn as nn

from oracle import painter_oracle as O

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from painter_amd import models_painter

ACCUM, ITERS, CLIP = 2, 4, 3.0


def _groups(named):
    named = list(named)
    return [{"params": [p for n, p in named if p.ndim > 1], "weight_decay": 0.05},
            {"params": [p for n, p in named if p.ndim <= 1], "weight_decay": 0.0}]


def _batches(cfg):
    out = []
    for it in range(ITERS):
        imgs, tgts, mask, valid = O.synthetic_batch(cfg, 2, 60 + it, "random")
        out.append((imgs, tgts, mask.reshape(2, *cfg.grid).to(torch.int32), valid))       # pairdataset.py:185-188: int mask [B, Hp, Wp]
    return out


def _reference_style_loop(model, optimizer, data, device, use_amp_context=True):
    """engine_train.train_one_epoch's body (see the module docstring); returns the per-iteration loss values and gradient norms."""
    scaler = torch.amp.GradScaler("cuda")
    model.train(True)
    optimizer.zero_grad()
    losses, norms = [], []
    for step, (samples, targets, bool_masked_pos, valid) in enumerate(data):
        samples = samples.to(device, non_blocking=True)
        targets = targets.to(device, non_blocking=True)
        bool_masked_pos = bool_masked_pos.to(device, non_blocking=True)
        valid = valid.to(device, non_blocking=True)
        with torch.autocast("cuda", enabled=use_amp_context):
            loss, y, mask = model(samples, targets, bool_masked_pos=bool_masked_pos, valid=valid)
        loss_value = loss.item()
        assert math.isfinite(loss_value)
        assert loss.dtype == torch.float32 and y.shape == (2, mask.shape[1], 16 * 16 * 3) and mask.dtype == torch.bool
        loss = loss / ACCUM
        update = (step + 1) % ACCUM == 0
        scaler.scale(loss).backward()
        norm = None
        if update:
            scaler.unscale_(optimizer)
            norm = torch.nn.utils.clip_grad_norm_(model.parameters(), CLIP)
            scaler.step(optimizer)
            scaler.update()
            optimizer.zero_grad()
        torch.cuda.synchronize()
        losses.append(loss_value)
        norms.append(None if norm is None else float(norm))
    return losses, norms


def _oracle_loop(P, cfg, data):
    names = list(P.keys())
    Pc = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    opt = torch.optim.AdamW(_groups([(n, Pc[n]) for n in names]), lr=2e-3, betas=(0.9, 0.95), eps=1e-3)
    opt.zero_grad()
    losses, norms = [], []
    for step, (imgs, tgts, mask, valid) in enumerate(data):
        lo, _, _ = O.forward(Pc, cfg, imgs, tgts, mask.reshape(2, -1), valid.clone())
        (lo / ACCUM).backward()
        norm = None
        if (step + 1) % ACCUM == 0:
            norm = float(torch.nn.utils.clip_grad_norm_([Pc[n] for n in names], CLIP))
            opt.step()
            opt.zero_grad()
        losses.append(lo.item())
        norms.append(norm)
    return Pc, losses, norms


def _module(cfg, P):
    m = models_painter.Painter(img_size=cfg.img_size, patch_size=cfg.patch_size, embed_dim=cfg.embed_dim, depth=cfg.depth,
                               num_heads=cfg.num_heads, drop_path_rate=0.0, window_size=14, qkv_bias=True, mlp_ratio=4,
                               norm_layer=partial(nn.LayerNorm, eps=1e-6), window_block_indexes=([0, 1], [3, 4]), residual_block_indexes=[],
                               use_rel_pos=True, out_feature="last_feat", decoder_embed_dim=cfg.decoder_embed_dim, loss_func="smoothl1",
                               compute_dtype="fp32")
    m.load_state_dict(P, strict=True)
    return m.cuda()


def _check(m, Pc, losses, norms, ref_losses, ref_norms):
    for a, b in zip(losses, ref_losses):
        assert abs(a - b) <= 2e-4 * abs(b), (losses, ref_losses)
    for a, b in zip(norms, ref_norms):
        assert (a is None) == (b is None)
        if a is not None:
            assert abs(a - b) <= 2e-3 * abs(b), (norms, ref_norms)
    worst = max(float((p.detach().cpu() - Pc[n].detach()).abs().max() / Pc[n].detach().abs().max().clamp_min(1e-6)) for n, p in m.named_parameters())
    assert worst < 1e-3, worst


def test_reference_training_loop_on_the_bare_module():
    cfg = O.small_config()
    P = O.random_params(cfg, 17)
    data = _batches(cfg)
    Pc, ref_losses, ref_norms = _oracle_loop(P, cfg, data)
    m = _module(cfg, P)
    opt = torch.optim.AdamW(_groups(m.named_parameters()), lr=2e-3, betas=(0.9, 0.95), eps=1e-3)
    losses, norms = _reference_style_loop(m, opt, data, torch.device("cuda"))
    _check(m, Pc, losses, norms, ref_losses, ref_norms)


def test_reference_training_loop_under_the_ddp_wrapper():
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel
    cfg = O.small_config()
    P = O.random_params(cfg, 17)
    data = _batches(cfg)
    Pc, ref_losses, ref_norms = _oracle_loop(P, cfg, data)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, world_size=1, rank=0)
    try:
        m = _module(cfg, P)
        ddp = DistributedDataParallel(m, device_ids=[0], find_unused_parameters=False)          # main_train.py:340
        opt = torch.optim.AdamW(_groups(ddp.module.named_parameters()), lr=2e-3, betas=(0.9, 0.95), eps=1e-3)
        losses, norms = _reference_style_loop(ddp, opt, data, torch.device("cuda"))
        assert ddp.module.patch_size == 16 and ddp.module.unpatchify is not None                   # engine_train.py:118,122
        _check(m, Pc, losses, norms, ref_losses, ref_norms)
    finally:
        dist.destroy_process_group()
