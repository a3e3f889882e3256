"""Where do the joules of a training step go?  ("in a power-capped regime, joules are the currency".)

The ViT-L B = 8 step runs at the board's power limit with the clock managed down (1.92 - 2.10 GHz by box), so a kernel family's cost to the
step is its ENERGY, not only its time: cycles saved without saving energy come back as a lower clock for everything else.  This probe loops
every kernel family of the step alone, whole-chip, for ~2 s each at the step's own shapes (R = 12544 token rows, D = 1024, 16 heads, 56 x 28
tokens), samples board power and shader clock every 100 ms (tools/power_probe.py's sampler), and prints per family

    ms per launch, board power and clock while it loops, joules per launch (above idle and total), joules per step (x launches per step),
    and joules per TFLOP of algorithmic work

next to the same for the whole step.  Launch counts per step: 27 block-units (24 blocks, blocks 0 - 2 on 2B rows).  Diagnostics only.

    python tools/energy_by_family.py [seconds per family]
"""
import sys
import time

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tools")
import bench  # noqa: E402
from painter_amd import models_painter, ops  # noqa: E402
from painter_amd._lib import EPI_BIAS, EPI_BIAS_RESID  # noqa: E402
from power_probe import Sampler  # noqa: E402


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
    dev = torch.device("cuda")
    s = Sampler()
    s.start()
    g = torch.Generator().manual_seed(0)
    T = torch.bfloat16
    R, D, Hd = 12544, 1024, 4096
    rnd = lambda *sh: ((torch.rand(sh, generator=g) * 2 - 1) * 0.5).to(T).to(dev)
    x, x4 = rnd(R, D), rnd(R, Hd)
    w_fc1, w_fc2, w_qkv, w_proj = rnd(Hd, D) * 0.1, rnd(D, Hd) * 0.1, rnd(3 * D, D) * 0.1, rnd(D, D) * 0.1
    b1, b4, b3 = torch.zeros(D, device=dev), torch.zeros(Hd, device=dev), torch.zeros(3 * D, device=dev)
    resid = torch.randn(R, D, generator=g).to(dev)
    out32 = torch.empty_like(resid)
    act, aux = ops.linear_gelu(x, w_fc1, b4)
    dy1, dy4, dy3 = rnd(R, D), rnd(R, Hd), rnd(R, 3 * D)
    L, H, Hp, Wp = 1568, 16, 56, 28
    qkv, dout = torch.randn(8 * L, 3 * D, generator=g).to(T).to(dev), torch.randn(8 * L, D, generator=g).to(T).to(dev)
    rel_h, rel_w = (torch.randn(111, 64, generator=g) * 0.05).to(dev), (torch.randn(55, 64, generator=g) * 0.05).to(dev)
    rcat, rcatT = ops.relpos_pack(rel_h, rel_w, Hp, Wp, T), ops.relpos_pack_t(rel_h, rel_w, Hp, Wp, T)
    ao, lse, tab = ops.attn_fwd(qkv, rcat, 8, L, H, Hp, Wp, 0.125, need_tables=True)
    gam, bet = torch.ones(D, device=dev), torch.zeros(D, device=dev)
    ln, mean, rstd = ops.layernorm_fwd(resid, gam, bet, 1e-6, T)
    dres = torch.randn(R, D, generator=g).to(dev)
    dxT = torch.empty(R, D, dtype=T, device=dev)
    fl_attn = 4.0 * 8 * H * L * L * 64          # forward FLOPs of one attention launch
    gf = lambda n, k: 2.0 * R * n * k

    # (name, fn, launches per step, algorithmic FLOPs per launch or 0, algorithmic bytes per launch or 0)
    fam = [
        ("qkv forward (bias)", lambda: ops.linear_fwd(x, w_qkv, b3, EPI_BIAS), 27, gf(3 * D, D), 0),
        ("proj forward (bias + residual, fp32)", lambda: ops.linear_fwd(x, w_proj, b1, EPI_BIAS_RESID, out=out32, resid=resid), 27, gf(D, D), 0),
        ("fc1 forward (bias + GELU + gelu' code)", lambda: ops.linear_gelu(x, w_fc1, b4), 27, gf(Hd, D), 0),
        ("fc2 forward (bias + residual, fp32)", lambda: ops.linear_fwd(x4, w_fc2, b1, EPI_BIAS_RESID, out=out32, resid=resid), 27, gf(D, Hd), 0),
        ("fc2 data gradient (x gelu', column sums)", lambda: ops.linear_dgrad(dy1, w_fc2, gelu_aux=aux, colsum_out=b4), 27, gf(Hd, D), 0),
        ("fc1 data gradient", lambda: ops.linear_dgrad(dy4, w_fc1), 27, gf(D, Hd), 0),
        ("proj data gradient", lambda: ops.linear_dgrad(dy1, w_proj), 27, gf(D, D), 0),
        ("qkv data gradient", lambda: ops.linear_dgrad(dy3, w_qkv), 27, gf(D, 3 * D), 0),
        ("fc1 weight gradient (+ slab sum)", lambda: ops.linear_wgrad(dy4, x), 54, gf(Hd, D), 0),        # fc1 and fc2: same FLOPs
        ("qkv weight gradient (+ slab sum)", lambda: ops.linear_wgrad(dy3, x), 27, gf(3 * D, D), 0),
        ("proj weight gradient (+ slab sum)", lambda: ops.linear_wgrad(dy1, x), 27, gf(D, D), 0),
        ("attention forward", lambda: ops.attn_fwd(qkv, rcat, 8, L, H, Hp, Wp, 0.125, need_tables=True), 27, fl_attn, 0),
        ("attention backward (dQ + dKV)", lambda: ops.attn_bwd_core(qkv, rcat, rcatT, ao, dout, lse, 8, L, H, Hp, Wp, 0.125, tables=tab), 27, 2 * fl_attn, 0),
        ("LayerNorm forward", lambda: ops.layernorm_fwd(resid, gam, bet, 1e-6, T, out=ln), 58, 0, 77e6),
        ("LayerNorm backward", lambda: ops.layernorm_bwd(dy1, resid, mean, rstd, gam, dres=dres, dx=dres, dxT=dxT), 58, 0, 205e6),
    ]
    idle = s.measure(lambda: time.sleep(0.05), 1.5, 0.0)
    m = models_painter.painter_vit_large_patch16_input896x448(compute_dtype="bf16")
    bench.randomize_parameters(m, seed=1)
    m = m.to(dev).train()
    c = m._cfg
    inp = bench.synthetic_inputs(8, c.H, c.W, c.L, 1234, dev)

    def step():
        for p in m.parameters():
            p.grad = None
        m._hot.relpos_stale()
        loss, _, _ = m(inp[0], inp[1], bool_masked_pos=inp[2], valid=inp[3])
        loss.backward()

    r = s.measure(step, 2 * secs, 0.0)
    s.card = max(s.nodes, key=lambda n_: r["per"][n_][0] - idle["per"][n_][0])
    p_idle = idle["per"][s.card][0]
    pm, _, cm, _ = r["per"][s.card]
    print("idle %.0f W; power cap %s uW" % (p_idle, s.cap[s.card]))
    print("%-44s %9s %7s %7s %9s %9s %9s %9s" % ("family (whole chip, alone, back to back)", "ms/launch", "W", "MHz", "J/launch", "x / step", "J / step", "J/TFLOP"))
    print("%-44s %9.3f %7.0f %7.0f %9.2f %9s %9.1f %9.1f   <- the whole step (two streams), %.1f J above idle"
          % ("training step, B = 8", r["ms"], pm, cm, pm * r["ms"] * 1e-3, "1", pm * r["ms"] * 1e-3, pm * r["ms"] * 1e-3 / (8 * 4.769), (pm - p_idle) * r["ms"] * 1e-3), flush=True)
    from painter_amd._lib import lib
    from painter_amd.engine import WGRAD_SIDE_TARGET as side_target
    lib.pa_debug_set(3, 0)                      # weight gradients sized for the whole chip while they run alone (the step sizes them for 96 workgroups beside the main stream)
    fam.append(("fc1 weight gradient at the side stream's sizing", None, 0, gf(Hd, D), 0))
    tot_j, tot_ms = 0.0, 0.0
    for name, fn, per_step, flops, nbytes in fam:
        if fn is None:
            lib.pa_debug_set(3, side_target)
            fn = lambda: ops.linear_wgrad(dy4, x)
        q = s.measure(fn, secs, 0.0)
        pw, _, ck, _ = q["per"][s.card]
        j = pw * q["ms"] * 1e-3
        tot_j += j * per_step
        tot_ms += q["ms"] * per_step
        eff = ("%9.1f" % (j / (flops / 1e12))) if flops else ("%6.1f/GB" % (j / (nbytes / 1e9)))
        print("%-44s %9.3f %7.0f %7.0f %9.3f %9d %9.1f %s" % (name, q["ms"], pw, ck, j, per_step, j * per_step, eff), flush=True)
    print("%-44s %9.2f %7s %7s %9s %9s %9.1f   (sum of the families above, run alone: %.1f ms of one-stream kernel time)"
          % ("sum", tot_ms, "", "", "", "", tot_j, tot_ms))
    s.alive = False


if __name__ == "__main__":
    main()
