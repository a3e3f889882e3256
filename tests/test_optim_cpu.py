"""CPU: the optimizer oracle (oracle/optim_oracle.py) pinned against torch.optim.AdamW + clip_grad_norm_ + the GradScaler skip
rule, i.e. against the very objects the reference uses (main_train.py:348, util/misc.py:256-268)."""
import torch

from oracle import optim_oracle as OO


def _setup(seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(7, 5), (33,), (4, 3, 2, 2), (129,)]
    params = [torch.randn(s, generator=g) for s in shapes]
    grads = [[torch.randn(s, generator=g) * (10.0 if k == 1 else 0.3) for s in shapes] for k in range(4)]
    cfg = [(1e-3, 0.05), (5e-4, 0.0), (1e-3 * 0.75, 0.05), (2e-4, 0.0)]          # (lr * lr_scale, weight decay) per tensor
    return params, grads, cfg


def test_oracle_matches_torch_adamw_with_unscale_and_clip():
    params, grads, cfg = _setup(0)
    ref_p = [torch.nn.Parameter(p.clone()) for p in params]
    opt = torch.optim.AdamW([{"params": [p], "lr": lr, "weight_decay": wd} for p, (lr, wd) in zip(ref_p, cfg)], lr=1e-3, betas=(0.9, 0.95))
    ora_p = [p.clone() for p in params]
    state = [dict(step=0, exp_avg=torch.zeros_like(p), exp_avg_sq=torch.zeros_like(p)) for p in params]
    scale, clip = 1024.0, 3.0
    for k in range(4):
        for p, g in zip(ref_p, grads[k]):
            p.grad = (g * scale).clone()
        # reference sequence: unscale_, clip_grad_norm_, step
        for p in ref_p:
            p.grad.div_(scale)
        ref_norm = torch.nn.utils.clip_grad_norm_(ref_p, clip)
        opt.step()
        norm, skipped = OO.scaled_clipped_step(ora_p, [g * scale for g in grads[k]], state, cfg, scale, clip, 0.9, 0.95, 1e-8)
        assert not skipped
        assert abs(float(norm) - float(ref_norm)) <= 1e-5 * float(ref_norm)
        for a, b in zip(ora_p, ref_p):
            assert torch.allclose(a, b.detach(), rtol=1e-6, atol=1e-7)
    for st, p in zip(state, ref_p):
        assert torch.allclose(st["exp_avg"], opt.state[p]["exp_avg"], rtol=1e-6, atol=1e-8)
        assert torch.allclose(st["exp_avg_sq"], opt.state[p]["exp_avg_sq"], rtol=1e-6, atol=1e-10)
        assert int(st["step"]) == int(opt.state[p]["step"])


def test_oracle_skips_on_non_finite_gradient():
    params, grads, cfg = _setup(1)
    ora_p = [p.clone() for p in params]
    state = [dict(step=0, exp_avg=torch.zeros_like(p), exp_avg_sq=torch.zeros_like(p)) for p in params]
    bad = [g.clone() for g in grads[0]]
    bad[2].view(-1)[3] = float("inf")
    _, skipped = OO.scaled_clipped_step(ora_p, bad, state, cfg, 1.0, 3.0)
    assert skipped and all(torch.equal(a, b) for a, b in zip(ora_p, params)) and all(st["step"] == 0 for st in state)
