"""Ablation as a profiler for the generation-3 attention kernels: runtime bit masks switch pieces of a kernel off (results are then
WRONG, only the time matters) and the difference is what the piece costs in place.
    python tools/attn_ablate.py              software-pipelined dQ (attn3s.hip, PA_ATTN3_ABL): staging, barrier, LDS fragment loads, exp, MFMA groups
    python tools/attn_ablate.py paired       paired 8-wave dQ (attn3p.hip, PA_ATTN3_ABL)
    python tools/attn_ablate.py epilogue     the default 4-wave kernels: dQ key loop / r-space step / stores (PA_ATTN3_DQ_ABL), dKV query loop
                                             (PA_ATTN3_DKV_ABL), forward key loop and table build (PA_ATTN3_FWD_ABL)
Times the dQ + dKV pair (dKV is constant, so differences are dQ's) or the forward."""
import os
import sys

import torch

sys.path.insert(0, ".")
from painter_amd import ops            # noqa: E402
from painter_amd._lib import lib       # noqa: E402


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
    B, H, Hp, Wp = 8, 16, 56, 28
    L = Hp * Wp
    g = torch.Generator().manual_seed(0)
    T = torch.bfloat16
    qkv = torch.randn(B * L, 3 * H * 64, generator=g).to(T).cuda()
    dout = torch.randn(B * L, H * 64, generator=g).to(T).cuda()
    rel_h = (torch.randn(2 * Hp - 1, 64, generator=g) * 0.05).cuda()
    rel_w = (torch.randn(2 * Wp - 1, 64, generator=g) * 0.05).cuda()
    rcat = ops.relpos_pack(rel_h, rel_w, Hp, Wp, T)
    rcatT = ops.relpos_pack_t(rel_h, rel_w, Hp, Wp, T)
    paired = len(sys.argv) > 1 and sys.argv[1] == "paired"
    if len(sys.argv) > 1 and sys.argv[1] == "epilogue":          # the default 4-wave dQ kernel: what is outside the key loop worth?
        lib.pa_attn_set_generation(0)
        out, lse, tables = ops.attn_fwd(qkv, rcat, B, L, H, Hp, Wp, 0.125, need_tables=True)
        names = {1: "no r-space loop", 2: "no dG stores", 4: "no dQ store", 16: "no key loop", 32: "no r-space gather", 64: "no r-space MFMA"}
        for _ in range(2):
            for m in (0, 1, 2, 32, 64, 2 + 32 + 64, 16, 16 + 1, 0):
                os.environ["PA_ATTN3_DQ_ABL"] = str(m)
                t = timeit(lambda: ops.attn_bwd_core(qkv, rcat, rcatT, out, dout, lse, B, L, H, Hp, Wp, 0.125, tables=tables))
                print("abl %3d  dq+dkv %.3f ms   [%s]" % (m, t, ", ".join(v for k, v in names.items() if m & k) or "full kernel"), flush=True)
        os.environ.pop("PA_ATTN3_DQ_ABL")
        for m in (0, 16, 0):
            os.environ["PA_ATTN3_DKV_ABL"] = str(m)
            t = timeit(lambda: ops.attn_bwd_core(qkv, rcat, rcatT, out, dout, lse, B, L, H, Hp, Wp, 0.125, tables=tables))
            print("dkv abl %3d  dq+dkv %.3f ms   [%s]" % (m, t, "no query loop in dKV" if m else "full kernel"), flush=True)
        os.environ.pop("PA_ATTN3_DKV_ABL")
        for m in (0, 16, 32, 48, 0):
            os.environ["PA_ATTN3_FWD_ABL"] = str(m)
            t = timeit(lambda: ops.attn_fwd(qkv, rcat, B, L, H, Hp, Wp, 0.125, need_tables=True))
            print("fwd abl %3d  forward %.3f ms   [%s]" % (m, t, ", ".join(v for k, v in {16: "no key loop", 32: "no table build"}.items() if m & k) or "full kernel"), flush=True)
        os.environ.pop("PA_ATTN3_FWD_ABL")
        return
    lib.pa_attn_set_generation(4 if paired else 5)
    out, lse, tables = ops.attn_fwd(qkv, rcat, B, L, H, Hp, Wp, 0.125, need_tables=True)
    if paired:      # attn3p.hip's mask
        names = {1: "no staging", 4: "no LDS frag loads", 8: "no exp / dS VALU", 16: "no write-back", 32: "no MFMAs", 128: "no phase barriers"}
        masks = [0, 1, 4, 8, 16, 32, 128, 8 + 16, 32 + 4, 1 + 4 + 8 + 16 + 32, 255, 0]
    else:
        names = {1: "no staging", 2: "no barrier", 4: "no LDS frag loads", 8: "no exp", 16: "no write-back", 32: "no MFMA group 2", 64: "no MFMA group 1"}
        masks = [0, 1, 3, 4, 7, 8, 16, 32, 64, 96, 96 + 8, 127, 0]
    for _ in range(2):
        for m in masks:
            os.environ["PA_ATTN3_ABL"] = str(m)
            t = timeit(lambda: ops.attn_bwd_core(qkv, rcat, rcatT, out, dout, lse, B, L, H, Hp, Wp, 0.125, tables=tables))
            print("abl %3d  dq+dkv %.3f ms   [%s]" % (m, t, ", ".join(v for k, v in names.items() if m & k) or "full kernel"), flush=True)
    os.environ.pop("PA_ATTN3_ABL")
    lib.pa_attn_set_generation(0)


if __name__ == "__main__":
    main()
