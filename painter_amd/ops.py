"""Thin tensor-level wrappers over the C ABI (include/painter_hip.h).  PyTorch is used only for device
memory (torch.empty), the current HIP stream and dtype tags; all arithmetic happens in libpainter_hip.so."""
import torch

from ._lib import (EPI_BIAS, EPI_BIAS_F32, EPI_BIAS_GELU, EPI_BIAS_RESID, PA_BF16, PA_F32, check, lib)

_WS = {}


def stream():
    return torch.cuda.current_stream().cuda_stream


def code(dtype):
    if dtype == torch.bfloat16:
        return PA_BF16
    if dtype == torch.float32:
        return PA_F32
    raise TypeError("painter_amd supports float32 / bfloat16 operand types, got %s" % dtype)


def p(t):
    return 0 if t is None else t.data_ptr()


def workspace(nbytes, device, slot=0):
    """Grow-only scratch per (device, slot); safe to reuse because every user is stream-ordered."""
    key = (device, slot)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _WS[key] = buf
    return buf


def _req(t, dtype=None):
    assert t.is_cuda and t.stride(-1) == 1, "device tensor with unit inner stride required"
    if dtype is not None:
        assert t.dtype == dtype, (t.dtype, dtype)
    return t


# ------------------------------------------------------------------------------------------- linear
def linear_fwd(x, w, bias, epilogue=EPI_BIAS, out=None, out2=None, resid=None, rowscale=None, rows_per_sample=1):
    """x [M,K] T (row stride free), w [N,K] T contiguous, bias [N] f32."""
    M, K = x.shape
    N = w.shape[0]
    T = x.dtype
    _req(x); _req(w, T)
    assert w.is_contiguous() and w.shape[1] == K
    if out is None:
        odt = torch.float32 if epilogue in (EPI_BIAS_F32, EPI_BIAS_RESID) else T
        out = torch.empty((M, N), dtype=odt, device=x.device)
    if epilogue == EPI_BIAS_RESID:
        assert resid is not None and resid.dtype == torch.float32 and resid.stride(0) == out.stride(0)
    check(lib.pa_linear_fwd(code(T), epilogue, p(x), x.stride(0), p(w), p(bias), p(out), p(out2), out.stride(0),
                            p(resid), p(rowscale), rows_per_sample, M, N, K, stream()), "pa_linear_fwd")
    return out


def linear_gelu(x, w, bias, need_pre=True):
    M = x.shape[0]
    N = w.shape[0]
    act = torch.empty((M, N), dtype=x.dtype, device=x.device)
    pre = torch.empty((M, N), dtype=x.dtype, device=x.device) if need_pre else None
    linear_fwd(x, w, bias, EPI_BIAS_GELU, out=act, out2=pre)
    return act, pre


def linear_pixshuf(x, w, bias, batch, Hp, Wp, P, C):
    T = x.dtype
    out = torch.empty((batch, Hp * P, Wp * P, C), dtype=T, device=x.device)
    check(lib.pa_linear_pixshuf(code(T), p(x), x.stride(0), p(w), p(bias), p(out), batch, Hp, Wp, P, C, x.shape[1],
                                stream()), "pa_linear_pixshuf")
    return out


def linear_dgrad(dy, w, pre=None, out=None):
    """dX[M,K] = dY[M,N] . W[N,K]  (* gelu'(pre) when pre is given)."""
    M, N = dy.shape
    K = w.shape[1]
    T = dy.dtype
    _req(dy); _req(w, T)
    if out is None:
        out = torch.empty((M, K), dtype=T, device=dy.device)
    if pre is not None:
        assert pre.stride(0) == out.stride(0)
    check(lib.pa_linear_dgrad(code(T), p(dy), dy.stride(0), p(w), p(pre), p(out), out.stride(0), M, N, K, stream()),
          "pa_linear_dgrad")
    return out


def linear_wgrad(dy, x, out=None):
    """dW[N,K] = dY[M,N]^T . X[M,K] in fp32."""
    M, N = dy.shape
    K = x.shape[1]
    T = dy.dtype
    _req(dy); _req(x, T)
    if out is None:
        out = torch.empty((N, K), dtype=torch.float32, device=dy.device)
    ws = workspace(lib.pa_linear_wgrad_workspace_bytes(code(T), M, N, K), dy.device)
    check(lib.pa_linear_wgrad(code(T), p(dy), dy.stride(0), p(x), x.stride(0), p(out), p(ws), M, N, K, stream()),
          "pa_linear_wgrad")
    return out


def colsum(x, out=None):
    M, N = x.shape
    if out is None:
        out = torch.empty((N,), dtype=torch.float32, device=x.device)
    ws = workspace(lib.pa_colsum_workspace_bytes(M, N), x.device)
    check(lib.pa_colsum(code(x.dtype), p(x), x.stride(0), M, N, p(out), p(ws), stream()), "pa_colsum")
    return out


# ------------------------------------------------------------------------------------------- layernorm
def layernorm_fwd(x, gamma, beta, eps, out_dtype, out=None):
    R, D = x.shape
    _req(x, torch.float32)
    if out is None:
        out = torch.empty((R, D), dtype=out_dtype, device=x.device)
    mean = torch.empty((R,), dtype=torch.float32, device=x.device)
    rstd = torch.empty((R,), dtype=torch.float32, device=x.device)
    check(lib.pa_layernorm_fwd(code(out.dtype), p(x), x.stride(0), p(gamma), p(beta), eps, p(out), out.stride(0),
                               p(mean), p(rstd), R, D, stream()), "pa_layernorm_fwd")
    return out, mean, rstd


def layernorm_bwd(dy, x, mean, rstd, gamma, dres=None, dx=None, dxT=None, rowscale=None, rows_per_sample=1):
    """-> dx (f32, = dres + LN'(dy)), dgamma_dbeta [2, D]."""
    R, D = x.shape
    if dx is None:
        dx = torch.empty((R, D), dtype=torch.float32, device=x.device)
    gb = torch.empty((2, D), dtype=torch.float32, device=x.device)
    ws = workspace(lib.pa_layernorm_bwd_workspace_bytes(R, D), x.device)
    check(lib.pa_layernorm_bwd(code(dy.dtype), p(dy), dy.stride(0), p(x), x.stride(0), p(mean), p(rstd), p(gamma),
                               p(dres), p(dx), dx.stride(0), p(dxT), 0 if dxT is None else dxT.stride(0), p(rowscale),
                               rows_per_sample, p(gb), p(ws), R, D, stream()), "pa_layernorm_bwd")
    return dx, gb


# ------------------------------------------------------------------------------------------- attention
def relpos_pack(rel_pos_h, rel_pos_w, Hp, Wp, dtype):
    nrp = lib.pa_relpos_rows_padded(Hp, Wp)
    rcat = torch.empty((nrp, 64), dtype=dtype, device=rel_pos_h.device)
    check(lib.pa_relpos_pack(code(dtype), p(rel_pos_h), p(rel_pos_w), p(rcat), Hp, Wp, stream()), "pa_relpos_pack")
    return rcat


def attn_fwd(qkv, rcat, batch, L, heads, Hp, Wp, scale):
    """qkv [batch*L, 3*heads*64] T -> (out [batch*L, heads*64] T, lse [batch*heads, L] f32)."""
    T = qkv.dtype
    out = torch.empty((batch * L, heads * 64), dtype=T, device=qkv.device)
    lse = torch.empty((batch * heads, L), dtype=torch.float32, device=qkv.device)
    check(lib.pa_attn_fwd(code(T), p(qkv), qkv.stride(0), p(rcat), p(out), out.stride(0), p(lse), batch, L, heads, Hp, Wp,
                          float(scale), stream()), "pa_attn_fwd")
    return out, lse
