"""CPU restatement of the reference's optimizer step -- TEST INFRASTRUCTURE ONLY (imported by tests/ and nothing else).

The reference issues, per update step (Painter/util/misc.py:256-268, Painter/engine_train.py:85-90):
    scaler.unscale_(optimizer)                              g <- g / scale; found_inf = any non-finite g
    norm = clip_grad_norm_(parameters, clip_grad)           g <- g * min(1, clip / (||g||_2 + 1e-6))      (misc.py:263)
    scaler.step(optimizer)                                  skipped when found_inf                         (misc.py:267)
on a torch.optim.AdamW (Painter/main_train.py:348; third-party: PyTorch, decoupled weight decay, no amsgrad) whose groups carry
lr = lr_sched * lr_scale (util/lr_sched.py:9-21, util/lr_decay.py:15-61).  AdamW's update is restated from its published
algorithm (torch/optim/adamw.py, _single_tensor_adamw).  Pinned in tests/test_optim_cpu.py against torch.optim.AdamW +
torch.nn.utils.clip_grad_norm_ themselves (they run on the CPU of any box).
"""
import math

import torch


def total_grad_norm(grads):
    """torch.nn.utils.clip_grad_norm_: 2-norm of the per-tensor 2-norms (util/misc.py:263, :288-300).  Accumulated in float64:
    the oracle is the exact value both torch's and the kernel's fp32 summation orders approximate (they differ from each other
    by ~1e-5 relative on 1 M-element tensors)."""
    return torch.norm(torch.stack([torch.norm(g.detach().double(), 2.0) for g in grads]), 2.0).float()


def adamw_update(p, g, m, v, step, lr, weight_decay, beta1, beta2, eps):
    """One AdamW update of fp32 tensors, in place; `step` is the 1-based count AFTER the increment."""
    p.mul_(1.0 - lr * weight_decay)
    m.lerp_(g, 1.0 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


def scaled_clipped_step(params, grads, state, groups, scale, clip_grad, beta1=0.9, beta2=0.999, eps=1e-8):
    """params/grads/state: parallel lists (state[i] = dict(step, exp_avg, exp_avg_sq)); groups[i] = (lr, weight_decay) of tensor i;
    grads are still multiplied by `scale`.  -> (unscaled pre-clip norm, skipped?)."""
    g32 = [g.float() / scale for g in grads]
    finite = all(bool(torch.isfinite(g).all()) for g in g32)
    norm = total_grad_norm(g32) if finite else torch.tensor(float("nan"))
    if not finite:
        return norm, True
    coef = 1.0
    if clip_grad is not None and clip_grad > 0:
        coef = min(1.0, float(clip_grad) / (float(norm) + 1e-6))
    for p, g, st, (lr, wd) in zip(params, g32, state, groups):
        st["step"] += 1
        adamw_update(p, g * coef, st["exp_avg"], st["exp_avg_sq"], st["step"], lr, wd, beta1, beta2, eps)
    return norm, False
