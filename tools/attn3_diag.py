"""Error breakdown of the generation-3 attention kernels against an fp64 reference, per output and per 32-row tile (where is it wrong?),
next to generation 2 on the same inputs.  Diagnostics only."""
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from painter_amd import ops            # noqa: E402
from painter_amd._lib import lib       # noqa: E402

DEV = "cuda"


def attn_reference(qkv, rel_h, rel_w, B, L, H, Hp, Wp, scale):
    D = H * 64
    x = qkv.double().view(B, L, 3, H, 64).permute(2, 0, 3, 1, 4).reshape(3, B * H, L, 64)
    q, k, v = x[0], x[1], x[2]
    attn = (q * scale) @ k.transpose(-2, -1)
    ih = (torch.arange(Hp)[:, None] - torch.arange(Hp)[None, :] + Hp - 1).to(qkv.device)
    iw = (torch.arange(Wp)[:, None] - torch.arange(Wp)[None, :] + Wp - 1).to(qkv.device)
    Rh, Rw = rel_h.double()[ih], rel_w.double()[iw]
    rq = q.reshape(B * H, Hp, Wp, 64)
    bh = torch.einsum("bhwc,hkc->bhwk", rq, Rh)
    bw = torch.einsum("bhwc,wkc->bhwk", rq, Rw)
    attn = (attn.view(B * H, Hp, Wp, Hp, Wp) + bh[..., :, None] + bw[..., None, :]).view(B * H, L, L)
    lse = torch.logsumexp(attn, dim=-1)
    o = attn.softmax(-1) @ v
    return o.view(B, H, L, 64).permute(0, 2, 1, 3).reshape(B * L, D), lse


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def fro(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def tile_err(a, b, rows):
    """max abs error per 32-row tile, relative to the global max of b"""
    a, b = a.double().reshape(rows, -1), b.double().reshape(rows, -1)
    e = (a - b).abs().max(dim=1).values / b.abs().max().clamp_min(1e-30)
    return e.view(-1, 32).max(dim=1).values


def main():
    shapes = [(1, 1, 8, 28), (1, 2, 16, 28), (2, 2, 56, 28)]
    for B, H, Hp, Wp in shapes:
        L = Hp * Wp
        g = torch.Generator().manual_seed(5)
        qkv = torch.randn(B * L, 3 * H * 64, generator=g).to(torch.bfloat16).to(DEV)
        dout = torch.randn(B * L, H * 64, generator=g).to(torch.bfloat16).to(DEV)
        rel_h = (torch.randn(2 * Hp - 1, 64, generator=g) * 0.2).to(DEV)
        rel_w = (torch.randn(2 * Wp - 1, 64, generator=g) * 0.2).to(DEV)
        rcat = ops.relpos_pack(rel_h, rel_w, Hp, Wp, torch.bfloat16)
        rcatT = ops.relpos_pack_t(rel_h, rel_w, Hp, Wp, torch.bfloat16)
        nh, nw = 2 * Hp - 1, 2 * Wp - 1
        q64 = qkv.double().clone().requires_grad_(True)
        rh64 = rcat[:nh].double().clone().requires_grad_(True)
        rw64 = rcat[nh:nh + nw].double().clone().requires_grad_(True)
        ref, lse_ref = attn_reference(q64, rh64, rw64, B, L, H, Hp, Wp, 0.125)
        ref.backward(dout.double())
        D = H * 64
        for gen_ in (2, 0, 5):
            lib.pa_attn_set_generation(gen_)
            out, lse, tables = ops.attn_fwd(qkv, rcat, B, L, H, Hp, Wp, 0.125, need_tables=True)
            dqkv, drcat = ops.attn_bwd(qkv, rcat, rcatT, out, dout, lse, B, L, H, Hp, Wp, 0.125, tables=tables)
            torch.cuda.synchronize()
            errs = dict(out=rel(out, ref.detach()), lse=rel(lse, lse_ref.detach()), dq=rel(dqkv[:, :D], q64.grad[:, :D]),
                        dk=rel(dqkv[:, D:2 * D], q64.grad[:, D:2 * D]), dv=rel(dqkv[:, 2 * D:], q64.grad[:, 2 * D:]),
                        drh=rel(drcat[:nh], rh64.grad), drw=rel(drcat[nh:nh + nw], rw64.grad))
            print("shape", (B, H, Hp, Wp), "gen", {0: "3 4-wave", 5: "3 pipelined dq", 4: "3 paired", 2: "2"}[gen_], {k: "%.2e" % v for k, v in errs.items()}, flush=True)
            fros = dict(out=fro(out, ref.detach()), dq=fro(dqkv[:, :D], q64.grad[:, :D]), dk=fro(dqkv[:, D:2 * D], q64.grad[:, D:2 * D]),
                        dv=fro(dqkv[:, 2 * D:], q64.grad[:, 2 * D:]), drh=fro(drcat[:nh], rh64.grad), drw=fro(drcat[nh:nh + nw], rw64.grad))
            print("      relative Frobenius", {k: "%.2e" % v for k, v in fros.items()}, flush=True)
            if gen_ != 2 and max(errs.values()) > 3e-2:
                torch.set_printoptions(precision=2, linewidth=250)
                print("  out tiles ", tile_err(out, ref.detach(), B * L)[:64])
                print("  dq  tiles ", tile_err(dqkv[:, :D].contiguous(), q64.grad[:, :D].contiguous(), B * L)[:64])
                print("  dk  tiles ", tile_err(dqkv[:, D:2 * D].contiguous(), q64.grad[:, D:2 * D].contiguous(), B * L)[:64])
                print("  dv  tiles ", tile_err(dqkv[:, 2 * D:].contiguous(), q64.grad[:, 2 * D:].contiguous(), B * L)[:64])
                print("  drh rows  ", ((drcat[:nh].double() - rh64.grad).abs().max(dim=1).values / rh64.grad.abs().max())[:40])
                print("  drw rows  ", ((drcat[nh:nh + nw].double() - rw64.grad).abs().max(dim=1).values / rw64.grad.abs().max())[:60])
                print("  lse head  ", (lse.double() - lse_ref.detach()).abs().view(B * H, L)[0, :64])
    lib.pa_attn_set_generation(0)


if __name__ == "__main__":
    main()
