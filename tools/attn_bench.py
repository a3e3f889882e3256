"""Times the fused attention forward / backward at the ViT-L shape (B' = 8 and 16, 16 heads, 56x28 tokens, hd 64) for the
generation-3 kernels -- round-3 arrangement (dG written for a gather GEMM, light workgroups interleaved) and round-4 default (rel-pos
table gradient contracted inside dQ, light workgroups dispatched last; the timed backward then INCLUDES the fixed-order sum of the
partials, the round-3 column includes the gather GEMM) -- and generation 2 (pa_attn_set_generation(2)), interleaved in one process.  Env knobs of the kernels apply (PA_ATTN3_FWD_STAGES, PA_ATTN3_DQ_WAVES, ...)."""
import sys

import torch

sys.path.insert(0, ".")
from painter_amd import ops   # noqa: E402
from painter_amd._lib import lib   # noqa: E402

DEV, T = "cuda", torch.bfloat16


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    H, Hp, Wp = 16, 56, 28
    L = Hp * Wp
    g = torch.Generator().manual_seed(0)
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    for B in (8, 16):
        qkv = (torch.randn(B * L, 3 * H * 64, generator=g)).to(T).to(DEV)
        dout = torch.randn(B * L, H * 64, generator=g).to(T).to(DEV)
        rel_h = (torch.randn(2 * Hp - 1, 64, generator=g) * 0.05).to(DEV)
        rel_w = (torch.randn(2 * Wp - 1, 64, generator=g) * 0.05).to(DEV)
        rcat = ops.relpos_pack(rel_h, rel_w, Hp, Wp, T)
        rcatT = ops.relpos_pack_t(rel_h, rel_w, Hp, Wp, T)
        fl = 4.0 * B * H * L * L * 64
        nrp = rcat.shape[0]
        cases = (("gen2", 2, 1, 1), ("gen3 round-3 arrangement", 0, 1, 1), ("gen3 fused", 0, 2, 1), ("gen3 fused + light-last", 0, 2, 2))
        res = {c[0]: [] for c in cases}
        for _ in range(rounds):
            for name, gen_, fuse, light in cases:
                lib.pa_attn_set_generation(gen_)
                lib.pa_debug_set(7, fuse)
                lib.pa_debug_set(8, light)
                out, lse, tables = ops.attn_fwd(qkv, rcat, B, L, H, Hp, Wp, 0.125, need_tables=True)
                tf = timeit(lambda: ops.attn_fwd(qkv, rcat, B, L, H, Hp, Wp, 0.125, need_tables=True))
                tb = timeit(lambda: ops.attn_bwd_core(qkv, rcat, rcatT, out, dout, lse, B, L, H, Hp, Wp, 0.125, tables=tables))

                def full():
                    dqkv, dg = ops.attn_bwd_core(qkv, rcat, rcatT, out, dout, lse, B, L, H, Hp, Wp, 0.125, tables=tables)
                    ops.attn_bwd_relpos(dg, qkv, nrp, B, L, H, Hp, Wp)
                tr = timeit(full)
                res[name].append((tf, tb, tr))
        lib.pa_attn_set_generation(0)
        lib.pa_debug_set(7, 0)
        lib.pa_debug_set(8, 0)
        for name, _, _, _ in cases:
            tf, tb, tr = (min(r[i] for r in res[name]) for i in range(3))
            print("B'=%d %-26s fwd %.3f ms (%.0f TFLOP/s)   bwd core %.3f ms (%.0f TFLOP/s algorithmic, 2.5x fwd)   bwd + rel-pos gradient %.3f ms   rounds fwd %s core %s full %s"
                  % (B, name, tf, fl / tf / 1e9, tb, 2.5 * fl / tb / 1e9, tr, ["%.3f" % r[0] for r in res[name]], ["%.3f" % r[1] for r in res[name]],
                     ["%.3f" % r[2] for r in res[name]]))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
