"""Coarse and per-phase s_memtime stamps of the paired (8-wave) dQ kernel, workgroup 0, waves 0 (group 0) and 4 (group 1).  Diagnostics."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from painter_amd import ops            # noqa: E402
from painter_amd._lib import lib       # noqa: E402


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
    lib.pa_attn_set_generation(4)
    out, lse, tables = ops.attn_fwd(qkv, rcat, B, L, H, Hp, Wp, 0.125, need_tables=True)
    for extra in (0, 1 + 4 + 8 + 16 + 32):
        os.environ["PA_ATTN3_ABL"] = str(512 + extra)
        for _ in range(3):
            ops.attn_bwd_core(qkv, rcat, rcatT, out, dout, lse, B, L, H, Hp, Wp, 0.125, tables=tables)
        torch.cuda.synchronize()
        buf = np.zeros(128, dtype=np.uint64)
        lib.pa_attn_trace_paired(buf.ctypes.data_as(ctypes.c_void_p))
        tr = buf.reshape(2, 64).astype(np.int64)
        print("ablation mask %d" % extra)
        for grp in range(2):
            c = tr[grp]
            print("  group %d: prologue %d  loop %d  tail+epilogue %d  (kernel %d cycles)" % (grp, c[1] - c[0], c[2] - c[1], c[3] - c[2], c[3] - c[0]))
            ph = c[8:56].reshape(12, 4)
            m = (ph[2:, 1] - ph[2:, 0]).mean()
            b1 = (ph[2:, 2] - ph[2:, 1]).mean()
            v = (ph[2:, 3] - ph[2:, 2]).mean()
            it = (ph[3:, 0] - ph[2:-1, 0]).mean()
            print("           per tile (tiles 2..11): M work %.0f | wait at M's barrier %.0f | V work %.0f | tile %.0f cycles" % (m, b1, v, it))
    os.environ.pop("PA_ATTN3_ABL")
    lib.pa_attn_set_generation(0)


if __name__ == "__main__":
    main()
