"""Round 6: which rounding separates the HIP bf16 attention backward from the reference's bf16-autocast one?

Hypothesis: Delta.  The reference's softmax backward (autocast keeps softmax in fp32) forms Delta_i = sum_j P_ij dP_ij from fp32 P, i.e.
from the UNROUNDED attention output; a flash backward (ours, the dQ prologue) forms Delta_i = rowsum(dO o O) from the bf16-ROUNDED O the
forward stored.  The rounding error of O (2^-9 relative per element) lands in Delta and from there COHERENTLY in every dS_ij = P_ij (dP_ij -
Delta_i) of the row: an error in dQ_i / dK that does not average out over the keys the way the per-element roundings of P and dS do.

Test, at the kernel level: the same backward launched twice on the same inputs -- (a) Delta taken by the kernel from the stored bf16 O
(the product route), (b) Delta handed in through the two-kernel route of pa_attn_bwd (its `delta` argument) computed from the float64
reference's O -- both against the float64 autograd gradients.  Diagnostics only.

    python tools/attn_delta_probe.py
"""
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tools")
from attn3_diag import attn_reference, fro, rel      # noqa: E402
from painter_amd import ops                          # noqa: E402
from painter_amd._lib import check, lib              # noqa: E402
from painter_amd.ops import code, p                  # noqa: E402

DEV = "cuda"


def bwd_with_delta(qkv, rcat, rcatT, out, dout, lse, delta, tables, B, L, H, Hp, Wp, scale):
    """pa_attn_bwd's two-kernel route: `delta` f32 [B * H, L] given, the tables' lse / Delta fields are filled from it."""
    T = qkv.dtype
    nrp, hd = rcat.shape
    dqkv = torch.empty_like(qkv)
    nb = lib.pa_attn_bwd_relpos_partials_bytes(code(T), B, L, H, Hp, Wp, hd)
    part = torch.empty((nb,), dtype=torch.uint8, device=qkv.device)
    aux = ops.workspace(lib.pa_attn_bwd_aux_bytes(B, L, H, Hp, Wp), qkv.device, slot=1)
    check(lib.pa_attn_bwd(code(T), p(qkv), qkv.stride(0), p(rcat), p(rcatT), p(dout), dout.stride(0), p(lse), p(delta), p(dqkv), p(None), p(part),
                          p(aux), p(tables), p(None), 0, B, L, H, Hp, Wp, hd, float(scale), ops.stream()), "pa_attn_bwd")
    return dqkv, ops.attn_bwd_relpos(part, qkv, nrp, B, L, H, Hp, Wp)


def main():
    torch.manual_seed(0)
    for B, H, Hp, Wp, ostd in ((2, 2, 56, 28, 1.0), (2, 2, 56, 28, 0.25), (1, 16, 56, 28, 1.0)):
        L, D = Hp * Wp, H * 64
        g = torch.Generator().manual_seed(5)
        qkv = torch.randn(B * L, 3 * D, generator=g)
        qkv[:, :2 * D] *= ostd                      # flatter softmax rows for the second case (P spread over many keys, as in early blocks)
        qkv = qkv.to(torch.bfloat16).to(DEV)
        dout = torch.randn(B * L, D, generator=g).to(torch.bfloat16).to(DEV)
        rel_h = (torch.randn(2 * Hp - 1, 64, generator=g) * 0.2).to(DEV)
        rel_w = (torch.randn(2 * Wp - 1, 64, generator=g) * 0.2).to(DEV)
        rcat = ops.relpos_pack(rel_h, rel_w, Hp, Wp, torch.bfloat16)
        rcatT = ops.relpos_pack_t(rel_h, rel_w, Hp, Wp, torch.bfloat16)
        nh, nw = 2 * Hp - 1, 2 * Wp - 1
        q64 = qkv.double().clone().requires_grad_(True)
        rh64 = rcat[:nh].double().clone().requires_grad_(True)
        rw64 = rcat[nh:nh + nw].double().clone().requires_grad_(True)
        ref, _ = attn_reference(q64, rh64, rw64, B, L, H, Hp, Wp, 0.125)
        ref.backward(dout.double())
        out, lse, tables = ops.attn_fwd(qkv, rcat, B, L, H, Hp, Wp, 0.125, need_tables=True)
        res = {}
        dq_a, dr_a = ops.attn_bwd(qkv, rcat, rcatT, out, dout, lse, B, L, H, Hp, Wp, 0.125, tables=tables)
        res["Delta from the stored bf16 O (product)"] = (dq_a, dr_a)
        # Delta from the float64 O: [B * L, D] -> per (sample, head, query)
        d64 = (ref.detach() * dout.double()).view(B, L, H, 64).sum(-1).permute(0, 2, 1).reshape(B * H, L).float().contiguous()
        res["Delta from the float64 O"] = bwd_with_delta(qkv, rcat, rcatT, out, dout, lse, d64, tables, B, L, H, Hp, Wp, 0.125)
        dbf = (out.double() * dout.double()).view(B, L, H, 64).sum(-1).permute(0, 2, 1).reshape(B * H, L).float().contiguous()
        res["Delta from bf16 O, two-kernel route (control)"] = bwd_with_delta(qkv, rcat, rcatT, out, dout, lse, dbf, tables, B, L, H, Hp, Wp, 0.125)
        torch.cuda.synchronize()
        print("shape B=%d H=%d %dx%d, q/k std %.2f" % (B, H, Hp, Wp, ostd))
        for name, (dqkv, drcat) in res.items():
            e = dict(dq=fro(dqkv[:, :D], q64.grad[:, :D]), dk=fro(dqkv[:, D:2 * D], q64.grad[:, D:2 * D]), dv=fro(dqkv[:, 2 * D:], q64.grad[:, 2 * D:]),
                     drh=fro(drcat[:nh], rh64.grad), drw=fro(drcat[nh:nh + nw], rw64.grad))
            m = dict(dq=rel(dqkv[:, :D], q64.grad[:, :D]), dk=rel(dqkv[:, D:2 * D], q64.grad[:, D:2 * D]))
            print("  %-48s rel Frobenius %s | rel-max %s" % (name, {k: "%.2e" % v for k, v in e.items()}, {k: "%.2e" % v for k, v in m.items()}), flush=True)


if __name__ == "__main__":
    main()
