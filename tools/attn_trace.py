"""Where does a tile iteration of the generation-3 dQ kernel spend its cycles?  Runs the traced build (pa_attn_trace) at the ViT-L
shape and prints, for two workgroups x waves 0 / 1, the mean s_memtime deltas between the seven stamps of an iteration:
  0 top | 1 operand fragments of group 1 arrived (explicit lgkmcnt(0) in the traced build) | 2 group-1 MFMAs issued, tr reads issued |
  3 VALU done (dS packed) | 4 group-2 MFMAs issued + write-back | 5 staging stores issued | 6 past the barrier"""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from painter_amd import ops            # noqa: E402
from painter_amd._lib import lib       # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    H, Hp, Wp = 16, 56, 28
    L = Hp * Wp
    g = torch.Generator().manual_seed(0)
    T = torch.bfloat16
    qkv = torch.randn(B * L, 3 * H * 64, generator=g).to(T).cuda()
    dout = torch.randn(B * L, H * 64, generator=g).to(T).cuda()
    rel_h = (torch.randn(2 * Hp - 1, 64, generator=g) * 0.05).cuda()
    rel_w = (torch.randn(2 * Wp - 1, 64, generator=g) * 0.05).cuda()
    rcat = ops.relpos_pack(rel_h, rel_w, Hp, Wp, T)
    rcatT = ops.relpos_pack_t(rel_h, rel_w, Hp, Wp, T)
    out, lse, tables = ops.attn_fwd(qkv, rcat, B, L, H, Hp, Wp, 0.125, need_tables=True)
    for _ in range(2):
        ops.attn_bwd_core(qkv, rcat, rcatT, out, dout, lse, B, L, H, Hp, Wp, 0.125, tables=tables)
    torch.cuda.synchronize()
    lib.pa_attn_trace(1, None)
    ops.attn_bwd_core(qkv, rcat, rcatT, out, dout, lse, B, L, H, Hp, Wp, 0.125, tables=tables)
    torch.cuda.synchronize()
    buf = np.zeros(2 * 2 * 64 * 8, dtype=np.uint64)
    lib.pa_attn_trace(0, buf.ctypes.data_as(ctypes.c_void_p))
    tr = buf.reshape(2, 2, 64, 8).astype(np.int64)
    names = ["top->frags", "frags->mfma1", "mfma1->valu", "valu->mfma2", "mfma2->stage", "stage->barrier", "iteration"]
    for wg in range(2):
        for wv in range(2):
            t = tr[wg, wv, 8:44]
            d = np.diff(t[:, :7], axis=1)
            it = t[1:, 0] - t[:-1, 0]
            print("wg %d wave %d  " % (wg, wv) + "  ".join("%s %6.0f" % (n, v) for n, v in zip(names[:6], d.mean(0))) + "   iteration %6.0f (min %d max %d)"
                  % (it.mean(), it.min(), it.max()))
    for wg in range(2):
        c = tr[wg, 0, 60:64, 0]
        print("wg %d wave 0 coarse: prologue %d  loop %d  epilogue %d  cycles (kernel start -> end %d)" % (wg, c[1] - c[0], c[2] - c[1], c[3] - c[2], c[3] - c[0]))
    print("first iterations of wg 1 wave 0:", (tr[1, 0, 1:9, 0] - tr[1, 0, 0:8, 0]).tolist())


if __name__ == "__main__":
    main()
