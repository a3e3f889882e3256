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
    """Grow-only scratch per (device, slot, current stream); safe to reuse because every user on one stream is ordered."""
    key = (device, slot, torch.cuda.current_stream(device).cuda_stream)
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


def linear_gelu(x, w, bias, need_aux=True):
    """-> (act = gelu(x W^T + b), aux): aux is what linear_dgrad(gelu_aux=...) needs of the pre-activation -- the pre-activation itself in the
    exact-fp32 build, the 8-bit code of gelu'(pre) (uint8 [M, N]; gelu_aux_decode) in the bf16 build (include/painter_hip.h, PA_EPI_BIAS_GELU)."""
    M = x.shape[0]
    N = w.shape[0]
    act = torch.empty((M, N), dtype=x.dtype, device=x.device)
    aux = torch.empty((M, N), dtype=gelu_aux_dtype(x.dtype), device=x.device) if need_aux else None
    assert aux is None or aux.stride(0) == act.stride(0)
    linear_fwd(x, w, bias, EPI_BIAS_GELU, out=act, out2=aux)
    return act, aux


G8_OFF, G8_RANGE = 0.13, 1.26


def gelu_aux_dtype(T):
    """dtype of linear_gelu's second result: the pre-activation (fp32 build), or the 8-bit code of gelu'(pre) (bf16 build, C ABI 6)."""
    return torch.float32 if T == torch.float32 else torch.uint8


def gelu_aux_decode(aux):
    """uint8 code -> gelu' (float32), as the fc2 data-gradient epilogue decodes it (include/painter_hip.h, PA_EPI_BIAS_GELU); tests / tools."""
    return aux.to(torch.float32) * (G8_RANGE / 255.0) - G8_OFF


def gelu_aux_encode(g):
    """gelu' values -> uint8 code (tests / tools)."""
    return torch.clamp(torch.floor((g.double() + G8_OFF) * (255.0 / G8_RANGE) + 0.5), 0, 255).to(torch.uint8)


def linear_pixshuf(x, w, bias, batch, Hp, Wp, P, C):
    T = x.dtype
    out = torch.empty((batch, Hp * P, Wp * P, C), dtype=T, device=x.device)
    check(lib.pa_linear_pixshuf(code(T), p(x), x.stride(0), p(w), p(bias), p(out), batch, Hp, Wp, P, C, x.shape[1],
                                stream()), "pa_linear_pixshuf")
    return out


def linear_dgrad(dy, w, gelu_aux=None, out=None, colsum_out=None):
    """dX[M,K] = dY[M,N] . W[N,K]  (* gelu'(pre) when gelu_aux, the second result of linear_gelu() in the same dtype, is given).
    colsum_out (f32 [K], optional): receives the column sums of dX as stored -- the bias gradient of the layer whose dY dX is -- from the
    GEMM's epilogue (bf16 fast path) instead of a separate pass."""
    M, N = dy.shape
    K = w.shape[1]
    T = dy.dtype
    _req(dy); _req(w, T)
    if out is None:
        out = torch.empty((M, K), dtype=T, device=dy.device)
    if gelu_aux is not None:      # (fp32: the pre-activation, ld in elements; bf16: the uint8 code, row pitch in bytes -- the same number either way)
        assert gelu_aux.stride(0) == out.stride(0) and gelu_aux.dtype == gelu_aux_dtype(T), (gelu_aux.dtype, gelu_aux.stride(0), out.stride(0))
    ws = None
    if colsum_out is not None:
        assert colsum_out.shape == (K,) and colsum_out.dtype == torch.float32 and colsum_out.is_contiguous()
        ws = workspace(lib.pa_linear_dgrad_workspace_bytes(M, K), dy.device, slot=2)
    check(lib.pa_linear_dgrad(code(T), p(dy), dy.stride(0), p(w), p(gelu_aux), p(out), out.stride(0), p(colsum_out), p(ws), M, N, K, stream()),
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


def layernorm_bwd_workspace_bytes(R, D):
    return int(lib.pa_layernorm_bwd_workspace_bytes(R, D))


def layernorm_bwd(dy, x, mean, rstd, gamma, dres=None, dx=None, dxT=None, rowscale=None, rows_per_sample=1, gb=None, dxT_colsum=None, defer=False, ws=None):
    """-> dx (f32, = dres + LN'(dy)), dgamma_dbeta [2, D] (written into `gb` when given).
    dxT_colsum (f32 [D], optional, needs dxT): receives the column sums of rowscale * dx -- the bias gradient of the nn.Linear whose dY
    dxT is -- from the same pass.
    defer=True: -> (dx, finish): the parameter-gradient partial rows stay in a private buffer and `finish()` (callable once, on any stream
    ordered behind this call) reduces them into gb / dxT_colsum and returns gb -- nothing downstream in the backward needs them.
    ws (defer only): a caller-owned uint8 buffer of layernorm_bwd_workspace_bytes(R, D) for the partial rows (engine.HotPath.ln_workspace:
    a ring reused across blocks and steps); a fresh tensor is allocated otherwise."""
    R, D = x.shape
    if dx is None:
        dx = torch.empty((R, D), dtype=torch.float32, device=x.device)
    if gb is None:
        gb = torch.empty((2, D), dtype=torch.float32, device=x.device)
    assert gb.shape == (2, D) and gb.is_contiguous() and gb.dtype == torch.float32
    nbytes = lib.pa_layernorm_bwd_workspace_bytes(R, D)
    if defer:
        if ws is None:
            ws = torch.empty((nbytes,), dtype=torch.uint8, device=x.device)
        assert ws.dtype == torch.uint8 and ws.numel() >= nbytes and ws.is_contiguous()
    else:
        ws = workspace(nbytes, x.device)
    if dxT_colsum is not None:
        assert dxT is not None and dxT_colsum.shape == (D,) and dxT_colsum.dtype == torch.float32 and dxT_colsum.is_contiguous()
    check(lib.pa_layernorm_bwd(code(dy.dtype), p(dy), dy.stride(0), p(x), x.stride(0), p(mean), p(rstd), p(gamma),
                               p(dres), p(dx), dx.stride(0), p(dxT), 0 if dxT is None else dxT.stride(0), p(rowscale),
                               rows_per_sample, 0 if defer else p(gb), p(dxT_colsum), p(ws), R, D, stream()), "pa_layernorm_bwd")
    if not defer:
        return dx, gb

    def finish():
        check(lib.pa_layernorm_bwd_reduce(p(ws), p(gb), p(dxT_colsum), int(dxT_colsum is not None), R, D, stream()), "pa_layernorm_bwd_reduce")
        return gb
    finish.buffer = ws
    return dx, finish


# ------------------------------------------------------------------------------------------- attention
def relpos_pack(rel_pos_h, rel_pos_w, Hp, Wp, dtype):
    """-> Rcat T [NRP, hd] = [rel_pos_h ; rel_pos_w ; 0] (hd = rel_pos_h.shape[1], the head dim)."""
    nrp = lib.pa_relpos_rows_padded(Hp, Wp)
    hd = rel_pos_h.shape[1]
    assert rel_pos_h.shape == (2 * Hp - 1, hd) and rel_pos_w.shape == (2 * Wp - 1, hd)
    rcat = torch.empty((nrp, hd), dtype=dtype, device=rel_pos_h.device)
    check(lib.pa_relpos_pack(code(dtype), p(rel_pos_h), p(rel_pos_w), p(rcat), Hp, Wp, hd, stream()), "pa_relpos_pack")
    return rcat


def relpos_pack_batch(tabs, nblocks, Hp, Wp, hd, dtype, rcat=None, rcatT=None):
    """Every block's Rcat [nblocks, NRP, hd] and Rcat^T [nblocks, hd, NRP] in one launch.  tabs: int64 device tensor of 2 * nblocks
    addresses (rel_pos_h of every block, then rel_pos_w of every block; fp32 [2Hp-1, hd] / [2Wp-1, hd] each)."""
    nrp = lib.pa_relpos_rows_padded(Hp, Wp)
    assert tabs.dtype == torch.int64 and tabs.numel() == 2 * nblocks and tabs.is_cuda and tabs.is_contiguous()
    if rcat is None:
        rcat = torch.empty((nblocks, nrp, hd), dtype=dtype, device=tabs.device)
    if rcatT is None:
        rcatT = torch.empty((nblocks, hd, nrp), dtype=dtype, device=tabs.device)
    assert rcat.shape == (nblocks, nrp, hd) and rcatT.shape == (nblocks, hd, nrp) and rcat.is_contiguous() and rcatT.is_contiguous()
    assert rcat.dtype == dtype and rcatT.dtype == dtype
    check(lib.pa_relpos_pack_batch(code(dtype), p(tabs), p(rcat), p(rcatT), nblocks, Hp, Wp, hd, stream()), "pa_relpos_pack_batch")
    return rcat, rcatT


def attn_fwd(qkv, rcat, batch, L, heads, Hp, Wp, scale, need_tables=False):
    """qkv [batch*L, 3*heads*hd] T -> (out [batch*L, heads*hd] T, lse [batch*heads, L] f32[, tables]).
    need_tables: also return the per-query bias tables the backward reuses (None when the kernels in use do not export them)."""
    T = qkv.dtype
    hd = rcat.shape[1]
    assert qkv.shape[1] == 3 * heads * hd
    out = torch.empty((batch * L, heads * hd), dtype=T, device=qkv.device)
    lse = torch.empty((batch * heads, L), dtype=torch.float32, device=qkv.device)
    tables = None
    if need_tables:
        nb = lib.pa_attn_tables_bytes(code(T), batch, L, heads, Hp, Wp, hd)
        if nb > 0:
            tables = torch.empty((nb,), dtype=torch.uint8, device=qkv.device)
    check(lib.pa_attn_fwd(code(T), p(qkv), qkv.stride(0), p(rcat), p(out), out.stride(0), p(lse), p(tables), batch, L, heads,
                          Hp, Wp, hd, float(scale), stream()), "pa_attn_fwd")
    return (out, lse, tables) if need_tables else (out, lse)


def attn_launch_counts():
    """-> {"fwd": (generic, generation 2, generation 3), "bwd": (...)}: host-side launch counts of pa_attn_fwd / pa_attn_bwd by kernel family
    since process start (tests assert with them which kernels a model configuration actually ran on)."""
    import ctypes
    buf = (ctypes.c_longlong * 6)()
    check(lib.pa_attn_launch_counts(ctypes.addressof(buf)), "pa_attn_launch_counts")
    v = [int(x) for x in buf]
    return {"fwd": tuple(v[:3]), "bwd": tuple(v[3:])}


def relpos_pack_t(rel_pos_h, rel_pos_w, Hp, Wp, dtype):
    nrp = lib.pa_relpos_rows_padded(Hp, Wp)
    hd = rel_pos_h.shape[1]
    rcatT = torch.empty((hd, nrp), dtype=dtype, device=rel_pos_h.device)
    check(lib.pa_relpos_pack_t(code(dtype), p(rel_pos_h), p(rel_pos_w), p(rcatT), Hp, Wp, hd, stream()), "pa_relpos_pack_t")
    return rcatT


def attn_bwd_core(qkv, rcat, rcatT, out, dout, lse, batch, L, heads, Hp, Wp, scale, tables=None, prep="fused"):
    """-> (dqkv T [batch*L, 3*heads*hd], dG): the data gradients and what attn_bwd_relpos() turns into the rel-pos table gradients --
    either the per-query bias gradients dG T [batch*L, heads*NRP], or (28-token-wide bf16 kernels with `tables`) the per-workgroup
    fp32 partial sums of the table gradient itself, a uint8 scratch tensor (the dQ kernel contracts them; dG never exists).
    tables: what attn_fwd(need_tables=True) returned (it carries the lse fields; the backward writes its delta field into it).
    prep (where the tables can carry Delta): "fused" = the dQ kernel computes Delta = rowsum(dO o O) itself (no extra launch: round 5),
    "launch" = the round-4 route, one pa_attn_bwd_prep launch in front (A/B, tests)."""
    T = qkv.dtype
    dev = qkv.device
    nrp, hd = rcat.shape
    delta = None
    o_arg = None
    if tables is not None and lib.pa_attn_bwd_prep_ok(code(T), L, Hp, Wp, hd):
        if prep == "fused":
            o_arg = out
        else:
            # Delta = rowsum(dO o O) goes straight into the table tiles together with the log-sum-exp fields: one launch instead of two
            check(lib.pa_attn_bwd_prep(code(T), p(out), out.stride(0), p(dout), dout.stride(0), p(lse), p(tables), batch, L, heads, Hp, Wp, hd,
                                       float(scale), stream()), "pa_attn_bwd_prep")
    else:
        delta = torch.empty((batch * heads, L), dtype=torch.float32, device=dev)
        check(lib.pa_attn_bwd_delta(code(T), p(out), out.stride(0), p(dout), dout.stride(0), p(delta), batch, L, heads, hd,
                                    stream()), "pa_attn_bwd_delta")
    dqkv = torch.empty_like(qkv)
    nb = lib.pa_attn_bwd_relpos_partials_bytes(code(T), batch, L, heads, Hp, Wp, hd) if tables is not None else 0
    dG = part = None
    if nb > 0:
        part = torch.empty((nb,), dtype=torch.uint8, device=dev)
    else:
        dG = torch.empty((batch * L, heads * nrp), dtype=T, device=dev)
    aux = workspace(lib.pa_attn_bwd_aux_bytes(batch, L, heads, Hp, Wp), dev, slot=1)
    check(lib.pa_attn_bwd(code(T), p(qkv), qkv.stride(0), p(rcat), p(rcatT), p(dout), dout.stride(0), p(lse), p(delta),
                          p(dqkv), p(dG), p(part), p(aux), p(tables), p(o_arg), 0 if o_arg is None else o_arg.stride(0), batch, L, heads, Hp, Wp, hd,
                          float(scale), stream()), "pa_attn_bwd")
    return dqkv, (dG if part is None else part)


def attn_bwd_relpos(dG, qkv, nrp, batch, L, heads, Hp, Wp, out=None):
    """-> drcat f32 [NRP, hd] = d[rel_pos_h ; rel_pos_w ; pad]: a parameter gradient (nothing downstream consumes it).
    dG: the second result of attn_bwd_core (per-query bias gradients -> gather GEMM; uint8 partials -> fixed-order sum)."""
    T = qkv.dtype
    hd = qkv.shape[1] // (3 * heads)
    drcat = out if out is not None else torch.empty((nrp, hd), dtype=torch.float32, device=qkv.device)
    assert drcat.shape == (nrp, hd) and drcat.is_contiguous()
    ws = workspace(lib.pa_attn_bwd_relpos_workspace_bytes(code(T), batch, L, heads, Hp, Wp, hd), qkv.device)
    if dG.dtype == torch.uint8:
        check(lib.pa_attn_bwd_relpos_reduce(p(dG), p(drcat), p(ws), batch, L, heads, Hp, Wp, hd, stream()), "pa_attn_bwd_relpos_reduce")
        return drcat
    check(lib.pa_attn_bwd_relpos(code(T), p(dG), p(qkv), qkv.stride(0), p(drcat), p(ws), batch, L, heads, Hp, Wp, hd,
                                 stream()), "pa_attn_bwd_relpos")
    return drcat


def attn_bwd(qkv, rcat, rcatT, out, dout, lse, batch, L, heads, Hp, Wp, scale, tables=None):
    """-> (dqkv T [batch*L, 3*heads*hd], drcat f32 [NRP, hd])."""
    dqkv, dG = attn_bwd_core(qkv, rcat, rcatT, out, dout, lse, batch, L, heads, Hp, Wp, scale, tables=tables)
    return dqkv, attn_bwd_relpos(dG, qkv, rcat.shape[0], batch, L, heads, Hp, Wp)


# ------------------------------------------------------------------------------------------- tokens / pos / merge
def cast_bf16(x, out=None):
    if out is None:
        out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    check(lib.pa_cast_bf16(p(x), p(out), x.numel(), stream()), "pa_cast_bf16")
    return out


def pos_fwd(op, pe, L, D):
    """op = (idx int32 [L, K], val f32 [L, K]) device tensors of hostmath.sparse_rows(M); pe f32 [S, D] -> pos f32 [L, D] = M . pe."""
    idx, val = op
    assert idx.dtype == torch.int32 and val.dtype == torch.float32 and idx.shape == val.shape == (L, idx.shape[1])
    pos = torch.empty((L, D), dtype=torch.float32, device=pe.device)
    check(lib.pa_pos_fwd(p(idx), p(val), idx.shape[1], p(pe), p(pos), L, D, stream()), "pa_pos_fwd")
    return pos


def pos_bwd(opT, gx, gy, dpe, S, D):
    """opT = sparse_rows(M^T) on the device; dpe f32 [S, D] <- M^T . (gx + gy)."""
    idx, val = opT
    assert idx.dtype == torch.int32 and val.dtype == torch.float32 and idx.shape == val.shape == (S, idx.shape[1])
    check(lib.pa_pos_bwd(p(idx), p(val), idx.shape[1], p(gx), p(gy), p(dpe), S, D, stream()), "pa_pos_bwd")


def patch_weight_pack(w, T, P, out=None):
    """conv weight f32 [D, 3, P, P] -> T [D, Kp] (Kp = 3*P*P rounded up to 8, zero padded): the operand pa_patch_embed_fwd takes."""
    D = w.shape[0]
    kp = (3 * P * P + 7) // 8 * 8
    if out is None:
        out = torch.empty((D, kp), dtype=T, device=w.device)
    check(lib.pa_patch_weight_pack(code(T), p(w), p(out), D, P, stream()), "pa_patch_weight_pack")
    return out


def patch_embed_fwd(T, imgs, tgts, w, bias, mask_token, seg_x, seg_y, pos, mask_u8, type_cls, type_ins, seg_type,
                    batch, Hp, Wp, P, D):
    """w: T [D, ldw] as patch_weight_pack returns it (a plain [D, 3*P*P] T copy is the same thing when P % 8 == 0)."""
    tokens = torch.empty((2 * batch * Hp * Wp, D), dtype=torch.float32, device=imgs.device)
    mbs = 0 if mask_u8.shape[0] == 1 else mask_u8.stride(0)
    assert w.dtype == T and w.shape[0] == D and w.stride(1) == 1
    check(lib.pa_patch_embed_fwd(code(T), p(imgs), p(tgts), p(w), w.stride(0), p(bias), p(mask_token), p(seg_x), p(seg_y), p(pos),
                                 p(mask_u8), mbs, p(type_cls), p(type_ins), p(seg_type), p(tokens), batch, Hp, Wp, P, D,
                                 stream()), "pa_patch_embed_fwd")
    return tokens


def patch_cols_ok(T, batch, L, P, D):
    """Does the bf16 im2col + gemm256 fast path of the patch embedding take this shape?"""
    return T == torch.bfloat16 and bool(lib.pa_patch_cols_ok(batch, L, P, D))


def patch_im2col(imgs, tgts, batch, Hp, Wp, P):
    """-> bf16 [2*B*L, 3*P*P]: the conv's im2col operand (rows = x-stream tokens then y-stream tokens, k = c*P*P + ph*P + pw)."""
    cols = torch.empty((2 * batch * Hp * Wp, 3 * P * P), dtype=torch.bfloat16, device=imgs.device)
    check(lib.pa_patch_im2col(p(imgs), p(tgts), p(cols), batch, Hp, Wp, P, stream()), "pa_patch_im2col")
    return cols


def patch_embed_fwd_cols(cols, w, bias, mask_token, seg_x, seg_y, pos, mask_u8, type_cls, type_ins, seg_type, batch, L, D):
    tokens = torch.empty((2 * batch * L, D), dtype=torch.float32, device=cols.device)
    mbs = 0 if mask_u8.shape[0] == 1 else mask_u8.stride(0)
    assert w.dtype == torch.bfloat16 and w.shape[0] == D and w.stride(1) == 1 and cols.is_contiguous()
    check(lib.pa_patch_embed_fwd_cols(p(cols), p(w), w.stride(0), p(bias), p(mask_token), p(seg_x), p(seg_y), p(pos), p(mask_u8), mbs,
                                      p(type_cls), p(type_ins), p(seg_type), p(tokens), batch, L, cols.shape[1], D, stream()), "pa_patch_embed_fwd_cols")
    return tokens


def patch_embed_wgrad(dpe, imgs, tgts, batch, Hp, Wp, P, D):
    dw = torch.empty((D, 3 * P * P), dtype=torch.float32, device=dpe.device)
    ws = workspace(lib.pa_patch_embed_wgrad_workspace_bytes(D, P), dpe.device)
    check(lib.pa_patch_embed_wgrad(code(dpe.dtype), p(dpe), p(imgs), p(tgts), p(dw), p(ws), batch, Hp, Wp, P, D, stream()),
          "pa_patch_embed_wgrad")
    return dw


def tokens_bwd(T, dx0, mask_u8, batch, L, D):
    dpe = torch.empty((2 * batch * L, D), dtype=T, device=dx0.device)
    sums = torch.empty((3, L, D), dtype=torch.float32, device=dx0.device)
    mbs = 0 if mask_u8.shape[0] == 1 else mask_u8.stride(0)
    check(lib.pa_tokens_bwd(code(T), p(dx0), p(mask_u8), mbs, p(dpe), p(sums), batch, L, D, stream()), "pa_tokens_bwd")
    return dpe, sums


def merge_fwd(x, rows_out, D):
    out = torch.empty((rows_out, D), dtype=torch.float32, device=x.device)
    check(lib.pa_merge_fwd(p(x), p(out), rows_out * D, stream()), "pa_merge_fwd")
    return out


def merge_bwd(T, dmerged, rowscale, rps, rows_half, D):
    dx = torch.empty((2 * rows_half, D), dtype=torch.float32, device=dmerged.device)
    dxT = torch.empty((2 * rows_half, D), dtype=T, device=dmerged.device)
    check(lib.pa_merge_bwd(code(T), p(dmerged), p(dx), p(dxT), p(rowscale), rps, rows_half, D, stream()), "pa_merge_bwd")
    return dx, dxT


def scale_cast(T, x, rowscale, rps, out=None):
    rows, D = x.shape
    if out is None:
        out = torch.empty((rows, D), dtype=T, device=x.device)
    check(lib.pa_scale_cast(code(T), p(x), p(out), p(rowscale), rps, rows, D, stream()), "pa_scale_cast")
    return out


def ensemble_resid(x0, a, batch, group, L, D):
    x1 = torch.empty_like(x0)
    check(lib.pa_ensemble_resid(p(x0), p(a), p(x1), batch, group, L, D, stream()), "pa_ensemble_resid")
    return x1


# ------------------------------------------------------------------------------------------- decoder tail / loss
def conv3x3_pack(w3, T):
    w3r = torch.empty((64, 9, 64), dtype=T, device=w3.device)
    wf = torch.empty((64, 9, 64), dtype=T, device=w3.device)
    check(lib.pa_conv3x3_pack(code(T), p(w3), p(w3r), p(wf), stream()), "pa_conv3x3_pack")
    return w3r, wf


def decoder_tail_fwd(x_nhwc, w3r, b3, gamma, beta, w1, b1, eps, save_y3=True):
    B, Hi, Wi, C = x_nhwc.shape
    T = x_nhwc.dtype
    y3 = torch.empty((B, Hi, Wi, C), dtype=T, device=x_nhwc.device) if save_y3 else None
    pred = torch.empty((B, 3, Hi, Wi), dtype=torch.float32, device=x_nhwc.device)
    check(lib.pa_decoder_tail_fwd(code(T), p(x_nhwc), p(w3r), p(b3), p(gamma), p(beta), p(w1), p(b1), p(y3), p(pred), B, Hi, Wi,
                                  float(eps), stream()), "pa_decoder_tail_fwd")
    return pred, y3


def decoder_tail_bwd_pointwise(dpred, y3, gamma, beta, w1, eps):
    B, Hi, Wi, C = y3.shape
    T = y3.dtype
    dy3 = torch.empty_like(y3)
    grads = torch.empty((324,), dtype=torch.float32, device=y3.device)
    ws = workspace(lib.pa_decoder_tail_bwd_workspace_bytes(B, Hi, Wi), y3.device)
    check(lib.pa_decoder_tail_bwd_pointwise(code(T), p(dpred), p(y3), p(gamma), p(beta), p(w1), p(dy3), p(grads), p(ws), B, Hi, Wi,
                                            float(eps), stream()), "pa_decoder_tail_bwd_pointwise")
    return dy3, grads


def conv3x3_dgrad_unshuffle(dy3, wf, batch, Hp, Wp, P):
    T = dy3.dtype
    dE = torch.empty((batch * Hp * Wp, P * P * 64), dtype=T, device=dy3.device)
    check(lib.pa_conv3x3_dgrad_unshuffle(code(T), p(dy3), p(wf), p(dE), batch, Hp, Wp, P, stream()), "pa_conv3x3_dgrad_unshuffle")
    return dE


def conv3x3_wgrad(dy3, x_nhwc):
    B, Hi, Wi, C = x_nhwc.shape
    dw = torch.empty((64, 64, 3, 3), dtype=torch.float32, device=dy3.device)
    ws = workspace(lib.pa_conv3x3_wgrad_workspace_bytes(B, Hi, Wi), dy3.device)
    check(lib.pa_conv3x3_wgrad(code(dy3.dtype), p(dy3), p(x_nhwc), p(dw), p(ws), B, Hi, Wi, stream()), "pa_conv3x3_wgrad")
    return dw


LOSS_KINDS = {"smoothl1": 0, "l1": 1, "l2": 2, "l1l2": 3}


def loss_fwd(pred, tgts, valid, mask_u8, P, ignore_rule, eps_den, kind, beta=0.01):
    B, _, Hi, Wi = pred.shape
    out = torch.empty((2,), dtype=torch.float32, device=pred.device)
    ws = workspace(lib.pa_loss_workspace_bytes(B, Hi, Wi), pred.device)
    mbs = 0 if mask_u8.shape[0] == 1 else mask_u8.stride(0)
    check(lib.pa_loss_fwd(p(pred), p(tgts), p(valid), p(mask_u8), mbs, p(out), p(ws), B, Hi, Wi, P, int(ignore_rule),
                          float(eps_den), LOSS_KINDS[kind], float(beta), stream()), "pa_loss_fwd")
    return out


def loss_bwd(pred, tgts, valid, mask_u8, dloss, loss_out, P, kind, beta=0.01):
    B, _, Hi, Wi = pred.shape
    dpred = torch.empty_like(pred)
    mbs = 0 if mask_u8.shape[0] == 1 else mask_u8.stride(0)
    check(lib.pa_loss_bwd(p(pred), p(tgts), p(valid), p(mask_u8), mbs, p(dloss), p(loss_out), p(dpred), B, Hi, Wi, P,
                          LOSS_KINDS[kind], float(beta), stream()), "pa_loss_bwd")
    return dpred


def patchify(pred, Hp, Wp, P):
    B = pred.shape[0]
    out = torch.empty((B, Hp * Wp, P * P * 3), dtype=torch.float32, device=pred.device)
    check(lib.pa_patchify(p(pred), p(out), B, Hp, Wp, P, stream()), "pa_patchify")
    return out
