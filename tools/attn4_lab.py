"""One timing of the attention backward core (ViT-L grid, B' = 8) for the key-loop ablation builds of the generation-4 dQ kernel
(library built with -DA4_ABLATION_BUILD, PA_ATTN4_DQ_LAB picks the variant per process; results are wrong unless it is 0).
    for lab in 0 1 2 4 8 16 18 32 33 63; do PA_ATTN4_DQ_LAB=$lab PAINTER_AMD_LIB=painter_amd/lib/libpainter_hip_abl.so python tools/attn4_lab.py; done"""
import os
import sys

import torch

sys.path.insert(0, ".")
from painter_amd import ops   # noqa: E402
from tools.attn_bench import timeit   # noqa: E402

DEV, T = "cuda", torch.bfloat16
B, H, Hp, Wp = 8, 16, 56, 28
L = Hp * Wp
g = torch.Generator().manual_seed(0)
qkv = torch.randn(B * L, 3 * H * 64, generator=g).to(T).to(DEV)
dout = torch.randn(B * L, H * 64, generator=g).to(T).to(DEV)
rel_h = (torch.randn(2 * Hp - 1, 64, generator=g) * 0.05).to(DEV)
rel_w = (torch.randn(2 * Wp - 1, 64, generator=g) * 0.05).to(DEV)
rcat = ops.relpos_pack(rel_h, rel_w, Hp, Wp, T)
rcatT = ops.relpos_pack_t(rel_h, rel_w, Hp, Wp, T)
out, lse, tables = ops.attn_fwd(qkv, rcat, B, L, H, Hp, Wp, 0.125, need_tables=True)
t = timeit(lambda: ops.attn_bwd_core(qkv, rcat, rcatT, out, dout, lse, B, L, H, Hp, Wp, 0.125, tables=tables), iters=10)
print("LAB %3s   bwd core %.3f ms" % (os.environ.get("PA_ATTN4_DQ_LAB", "0"), t), flush=True)
if os.environ.get("PA_ATTN4_DQ_LAB") == "64":
    import ctypes
    import numpy as np
    from painter_amd._lib import lib
    torch.cuda.synchronize()
    buf = np.zeros(64 * 8, dtype=np.uint64)
    lib.pa_attn4_trace(buf.ctypes.data_as(ctypes.c_void_p))
    t = buf.reshape(64, 8).astype(np.int64)
    names = ["top (stores, loads, window)", "G1 (A: S/dP + tr reads)", "G2 (B: S/dP | A softmax)", "G3 (A: dQ/E | B softmax)", "barrier", "G4 (B: dQ/E | next frags)", "to next iteration (window write-back)"]
    for j in (2, 3, 10, 20, 30, 40):
        d = [int(t[j][k + 1] - t[j][k]) for k in range(6)] + [int(t[j + 1][0] - t[j][6])]
        print("iteration %2d: total %5d cycles  " % (j, int(t[j + 1][0] - t[j][0])) + "  ".join("%s %d" % (n.split(" ")[0], v) for n, v in zip(names, d)))
