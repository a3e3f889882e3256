"""Where a generation-4 dQ launch spends its time: the launch with parts switched off (PA_ATTN4_DQ_ABL is read per launch; results are
wrong with a bit set, only the time matters).  ViT-L grid, B' = 8.  python tools/attn4_ablate.py"""
import os
import sys

import torch

sys.path.insert(0, ".")
from painter_amd import ops   # noqa: E402
from painter_amd._lib import lib   # noqa: E402
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
for name, env in (("full", {}), ("no key loop", {"PA_ATTN4_DQ_ABL": "16"}), ("no key loop, no r-space", {"PA_ATTN4_DQ_ABL": "17"}),
                  ("no key loop, no r-space, no contraction", {"PA_ATTN4_DQ_ABL": "19"}), ("no r-space, no contraction", {"PA_ATTN4_DQ_ABL": "3"}),
                  ("gen3 only (knob 9 = 1)", {"_g3": "1"})):
    for k in ("PA_ATTN4_DQ_ABL",):
        os.environ.pop(k, None)
    lib.pa_debug_set(9, 1 if "_g3" in env else 0)
    for k, v in env.items():
        if not k.startswith("_"):
            os.environ[k] = v
    t = timeit(lambda: ops.attn_bwd_core(qkv, rcat, rcatT, out, dout, lse, B, L, H, Hp, Wp, 0.125, tables=tables), iters=10)
    print("%-44s bwd core %.3f ms" % (name, t), flush=True)
