"""Ablation as a profiler for the generation-3 attention kernels: runtime bit masks switch pieces of a kernel off (results are then
WRONG, only the time matters) and the difference is what the piece costs in place.
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
    raise SystemExit("usage: python tools/attn_ablate.py epilogue   (the paired / software-pipelined builds were retired: tools/experiments/)")


if __name__ == "__main__":
    main()
