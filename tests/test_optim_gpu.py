"""GPU parity of the fused optimizer step (csrc/optim.hip through painter_amd/optim.py) against the CPU oracle, which is itself
pinned to torch.optim.AdamW + clip_grad_norm_ (tests/test_optim_cpu.py).  fp32 elementwise arithmetic in the same operation
order: gate 2e-6 relative (fma contraction / reciprocal forms differ in the last bit)."""
import pytest
import torch

from oracle import optim_oracle as OO

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from painter_amd import optim as PO

SHAPES = [(1024, 1024), (3072,), (5, 7), (64, 64, 3, 3), (1, 197, 64), (8193,), (3,)]


def _make(seed):
    g = torch.Generator().manual_seed(seed)
    params = [torch.randn(s, generator=g) for s in SHAPES]
    cfg = [(1e-3 * (0.75 ** (i % 3)), 0.05 if len(s) > 1 else 0.0) for i, s in enumerate(SHAPES)]
    return g, params, cfg


@pytest.mark.parametrize("scale,clip", [(65536.0, 3.0), (1.0, None), (128.0, 0.05)])
def test_fused_adamw_vs_oracle(scale, clip):
    g, params, cfg = _make(3)
    dev_p = [torch.nn.Parameter(p.clone().cuda()) for p in params]
    opt = PO.AdamW([{"params": [p], "lr": lr, "weight_decay": wd} for p, (lr, wd) in zip(dev_p, cfg)], lr=1e-3, betas=(0.9, 0.95))
    ora_p = [p.clone() for p in params]
    state = [dict(step=0, exp_avg=torch.zeros_like(p), exp_avg_sq=torch.zeros_like(p)) for p in params]
    scale_t = torch.tensor(scale, device="cuda")
    for k in range(3):
        grads = [torch.randn(s, generator=g) * (5.0 if k == 1 else 0.2) * scale for s in SHAPES]
        for p, gr in zip(dev_p, grads):
            p.grad = gr.clone().cuda()
        info = opt.grad_sumsq()
        opt.step(grad_scale=scale_t, max_norm=clip)
        norm, skipped = OO.scaled_clipped_step(ora_p, grads, state, cfg, scale, clip, 0.9, 0.95, 1e-8)
        assert not skipped and float(info[1]) == 0.0
        dev_norm = float(torch.sqrt(info[0])) / scale
        assert abs(dev_norm - float(norm)) <= 2e-6 * float(norm)          # fp32 per-chunk sums, float64 across chunks
        for a, b in zip(dev_p, ora_p):
            assert torch.allclose(a.detach().cpu(), b, rtol=2e-6, atol=1e-7)
    for p, st in zip(dev_p, state):
        assert torch.allclose(opt.state[p]["exp_avg"].cpu(), st["exp_avg"], rtol=2e-6, atol=1e-7)      # cancellation in m + (g - m)(1 - b1)
        assert torch.allclose(opt.state[p]["exp_avg_sq"].cpu(), st["exp_avg_sq"], rtol=2e-6, atol=1e-9)
        assert float(opt.state[p]["step"]) == float(st["step"])


def test_fused_step_skips_on_inf_and_scaler_backs_off():
    g, params, cfg = _make(4)
    dev_p = [torch.nn.Parameter(p.clone().cuda()) for p in params]
    opt = PO.AdamW([{"params": [p], "lr": lr, "weight_decay": wd} for p, (lr, wd) in zip(dev_p, cfg)], lr=1e-3)
    scaler = PO.NativeScalerWithGradNormCount(init_scale=1024.0)
    # a scalar "loss" whose gradient w.r.t. every parameter is known: sum(w_i * p_i)
    ws = [torch.randn(s, generator=g).cuda() for s in SHAPES]
    loss = sum((w * p).sum() for w, p in zip(ws, dev_p))
    norm = scaler(loss, opt, clip_grad=3.0, parameters=dev_p)
    ref_norm = torch.sqrt(sum((w.double() ** 2).sum() for w in ws))
    assert abs(float(norm) - float(ref_norm)) < 1e-5 * float(ref_norm)
    assert float(scaler.state_dict()["scale"]) == 1024.0 and float(opt.state[dev_p[0]]["step"]) == 1.0
    before = [p.detach().clone() for p in dev_p]
    opt.zero_grad()
    ws[2].view(-1)[1] = float("nan")
    loss = sum((w * p).sum() for w, p in zip(ws, dev_p))
    scaler(loss, opt, clip_grad=3.0, parameters=dev_p)
    assert all(torch.equal(a, b.detach()) for a, b in zip(before, dev_p))            # step skipped on the device
    assert float(opt.state[dev_p[0]]["step"]) == 1.0                                  # ... and not counted
    assert float(scaler.state_dict()["scale"]) == 512.0                               # GradScaler back-off


def test_torch_gradscaler_drives_fused_adamw_and_state_dict_roundtrip():
    """The unchanged reference scaler (util/misc.py:252-270) on top of our optimizer, and torch.optim.AdamW checkpoint interchange."""
    g, params, cfg = _make(5)
    dev_p = [torch.nn.Parameter(p.clone().cuda()) for p in params]
    ref_p = [torch.nn.Parameter(p.clone().cuda()) for p in params]
    groups = lambda ps: [{"params": [p], "lr": lr, "weight_decay": wd} for p, (lr, wd) in zip(ps, cfg)]
    opt, ref = PO.AdamW(groups(dev_p), lr=1e-3, betas=(0.9, 0.95)), torch.optim.AdamW(groups(ref_p), lr=1e-3, betas=(0.9, 0.95))
    s1, s2 = torch.amp.GradScaler("cuda", init_scale=256.0), torch.amp.GradScaler("cuda", init_scale=256.0)
    ws = [torch.randn(s, generator=g).cuda() for s in SHAPES]
    for it in range(2):
        for ps, o, sc in ((dev_p, opt, s1), (ref_p, ref, s2)):
            o.zero_grad()
            loss = sum((w * p * p).sum() for w, p in zip(ws, ps))
            sc.scale(loss).backward()
            sc.unscale_(o)
            torch.nn.utils.clip_grad_norm_(ps, 3.0)
            sc.step(o)
            sc.update()
        for a, b in zip(dev_p, ref_p):
            assert torch.allclose(a.detach(), b.detach(), rtol=2e-6, atol=1e-7)
    # checkpoint written by torch.optim.AdamW loads into ours (and continues identically)
    opt2 = PO.AdamW(groups(dev_p), lr=1e-3, betas=(0.9, 0.95))
    import copy
    opt2.load_state_dict(copy.deepcopy(ref.state_dict()))     # state_dict() hands out the live tensors; a checkpoint file would not
    for ps, o in ((dev_p, opt2), (ref_p, ref)):
        o.zero_grad()
        sum((w * p * p).sum() for w, p in zip(ws, ps)).backward()
        o.step()
    for a, b in zip(dev_p, ref_p):
        assert torch.allclose(a.detach(), b.detach(), rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize("bind", [False, True])
def test_fused_step_invalidates_the_engines_bf16_weight_cache(bind):
    """The model keeps bf16 copies of its weight matrices keyed on the parameter version: after a fused step (which rewrites the
    parameters from a HIP kernel) the next forward must use the new weights -- same loss as a fresh model holding them."""
    from functools import partial

    import torch.nn as nn

    from oracle import painter_oracle as O
    from painter_amd import models_painter
    cfg = O.small_config()

    def make():
        m = models_painter.Painter(img_size=cfg.img_size, patch_size=16, embed_dim=cfg.embed_dim, depth=cfg.depth, num_heads=cfg.num_heads,
                                   drop_path_rate=0.0, mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6), use_rel_pos=True,
                                   decoder_embed_dim=cfg.decoder_embed_dim, compute_dtype="bf16")
        m.load_state_dict(O.random_params(cfg, 7), strict=True)
        return m.cuda().eval()

    imgs, tgts, mask, valid = O.synthetic_batch(cfg, 2, 8, "random")
    args = (imgs.cuda(), tgts.cuda())
    kw = lambda: dict(bool_masked_pos=mask.cuda(), valid=valid.clone().cuda())
    m = make()
    opt = PO.AdamW(m.parameters(), lr=1e-2, weight_decay=0.0)
    if bind:
        opt.bind_model(m)                               # the update pass also rewrites the cached bf16 copies
    loss0, _, _ = m(*args, **kw())
    loss0.backward()
    opt.step()
    loss1, _, _ = m(*args, **kw())                      # must see the updated weights
    m2 = make()
    m2.load_state_dict(m.state_dict())
    loss2, _, _ = m2(*args, **kw())
    assert abs(loss1.item() - loss2.item()) <= 1e-6 * abs(loss2.item()), (loss0.item(), loss1.item(), loss2.item())
    assert abs(loss1.item() - loss0.item()) > 1e-4 * abs(loss0.item())          # and the step did change something


def test_state_dict_of_multi_parameter_groups_loads_into_torch_adamw():
    """Our state_dict() -> torch.optim.AdamW.load_state_dict(): every parameter carries its OWN 0-d `step` tensor (torch's foreach
    step increments them in place: a tensor shared by the parameters of a group would be advanced once per parameter)."""
    import copy
    g, params, _ = _make(9)
    mk = lambda: [torch.nn.Parameter(p.clone().cuda()) for p in params]
    ours_p, torch_p = mk(), mk()
    groups = lambda ps: [{"params": [p for p in ps if p.ndim > 1], "weight_decay": 0.05}, {"params": [p for p in ps if p.ndim <= 1], "weight_decay": 0.0}]
    ours = PO.AdamW(groups(ours_p), lr=2e-3, betas=(0.9, 0.95))
    ws = [torch.randn(s, generator=g).cuda() for s in SHAPES]

    def one_step(ps, o):
        o.zero_grad()
        sum((w * p * p).sum() for w, p in zip(ws, ps)).backward()
        o.step()

    for _ in range(3):
        one_step(ours_p, ours)
    sd = ours.state_dict()
    steps = [st["step"] for st in sd["state"].values()]
    assert len({t.data_ptr() for t in steps}) == len(steps) and all(float(t) == 3.0 for t in steps)
    ref = torch.optim.AdamW(groups(torch_p), lr=2e-3, betas=(0.9, 0.95))
    with torch.no_grad():
        for a, b in zip(torch_p, ours_p):
            a.copy_(b)
    ref.load_state_dict(copy.deepcopy(sd))
    for _ in range(2):                                           # both continue from step 3
        one_step(ours_p, ours)
        one_step(torch_p, ref)
    assert all(float(st["step"]) == 5.0 for st in ref.state_dict()["state"].values())
    for a, b in zip(ours_p, torch_p):
        assert torch.allclose(a.detach(), b.detach(), rtol=2e-6, atol=1e-7)


def test_step_is_skipped_when_the_sum_of_squares_overflows():
    """Finite gradients whose fp32 sum of squares is inf (norm = inf): torch's clip would scale by 0 / produce NaN; the fused step must
    leave parameters and moments untouched, as it does for an inf gradient."""
    p = torch.nn.Parameter(torch.ones(4096, device="cuda"))
    opt = PO.AdamW([p], lr=1e-2)
    p.grad = torch.full_like(p, 3e19)                            # (3e19)^2 * 4096 > fp32 max
    before = p.detach().clone()
    opt.step(max_norm=1.0)
    assert torch.equal(p.detach(), before)
    p.grad = torch.full_like(p, 1.0)
    opt.step(max_norm=1.0)
    assert not torch.equal(p.detach(), before)


def test_gradient_buffers_replaced_between_steps_are_picked_up():
    """`p.grad = new_tensor` between steps (what autograd does after zero_grad(set_to_none=True)) must not leave the launcher's cached
    pointer table aimed at the old buffers."""
    g, params, cfg = _make(21)
    ours_p = [torch.nn.Parameter(p.clone().cuda()) for p in params]
    ref_p = [torch.nn.Parameter(p.clone().cuda()) for p in params]
    ours, ref = PO.AdamW(ours_p, lr=1e-3), torch.optim.AdamW(ref_p, lr=1e-3)
    keep = []
    for it in range(3):
        grads = [torch.randn(s, generator=g).cuda() for s in SHAPES]
        keep.append(grads)                                       # old buffers stay alive, so a stale pointer would read valid but wrong data
        for p, q, gr in zip(ours_p, ref_p, grads):
            p.grad, q.grad = gr.clone(), gr.clone()
        ours.step()
        ref.step()
    for a, b in zip(ours_p, ref_p):
        assert torch.allclose(a.detach(), b.detach(), rtol=2e-6, atol=1e-7)


def test_plain_step_after_a_gradscaler_step():
    """torch's GradScaler.step() deletes `optimizer.grad_scale` / `found_inf` after its call; a later step without the scaler must
    still work (class-level defaults), and must not re-use the scaler's factors."""
    p = torch.nn.Parameter(torch.ones(257, device="cuda"))
    q = torch.nn.Parameter(torch.ones(257, device="cuda"))
    ours, ref = PO.AdamW([p], lr=1e-2), torch.optim.AdamW([q], lr=1e-2)
    s1, s2 = torch.amp.GradScaler("cuda", init_scale=64.0), torch.amp.GradScaler("cuda", init_scale=64.0)
    for par, opt, sc in ((p, ours, s1), (q, ref, s2)):
        sc.scale((par * par).sum()).backward()
        sc.step(opt)
        sc.update()
        opt.zero_grad()
        (par * par * 3.0).sum().backward()
        opt.step()                                               # no scaler
    assert torch.allclose(p.detach(), q.detach(), rtol=2e-6, atol=1e-7)
