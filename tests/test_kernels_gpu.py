"""GPU parity tests of the individual HIP kernels (through the C ABI) against plain PyTorch fp32/fp64
references of the same op.  fp32 build: gate 1e-4..1e-3 (exact-fp32 MFMA); bf16 build: gate relative to the
bf16 rounding of the operands (stated per test)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from painter_amd import ops
    from painter_amd._lib import EPI_BIAS, EPI_BIAS_F32, EPI_BIAS_GELU, EPI_BIAS_RESID

DEV = "cuda"


def relerr(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def gen(shape, seed, scale=1.0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype).to(DEV)


TOL = {torch.float32: 2e-5, torch.bfloat16: 2e-2}


def gelu_grad_ref(pre):
    """d gelu / d pre (erf GELU, nn.GELU's default: models_painter.py:253) in float64 through torch autograd."""
    pg = pre.double().clone().requires_grad_(True)
    torch.nn.functional.gelu(pg).sum().backward()
    return pg.grad


def gelu_aux_of(pre):
    """What ops.linear_gelu hands to ops.linear_dgrad(gelu_aux=...) for a pre-activation tensor of this dtype: the pre-activation itself
    (fp32 build), the 8-bit code of gelu'(pre) (bf16 build since C ABI 6: q = round((g' + 0.13) * 255 / 1.26); include/painter_hip.h, PA_EPI_BIAS_GELU)."""
    return pre if pre.dtype == torch.float32 else ops.gelu_aux_encode(gelu_grad_ref(pre))


def gelu_aux_values(aux):
    """The derivative values the fc2 data-gradient epilogue multiplies with: float64 of the decoded 8-bit code (bf16 build: C ABI 6), or
    erf-GELU' of the saved pre-activation (fp32 build)."""
    return ops.gelu_aux_decode(aux).double() if aux.dtype == torch.uint8 else gelu_grad_ref(aux)


def check_gelu_pair(x, w, b, act, aux, tol):
    """act / aux of ops.linear_gelu against the pre-activation the same kernel rounds (the bias epilogue on the same operands: the same
    accumulation order, hence the same bits in front of the GELU)."""
    pre = ops.linear_fwd(x, w, b, EPI_BIAS)
    assert relerr(act.float(), torch.nn.functional.gelu(pre.double())) < tol
    if pre.dtype == torch.float32:
        assert torch.equal(aux, pre)
    else:     # the 8-bit code: absolute error <= half a step (2.5e-3) + the 1.5e-7 of the erf approximation, against max |gelu'| = 1.13
        assert aux.dtype == torch.uint8 and aux.shape == pre.shape
        assert float((ops.gelu_aux_decode(aux).double() - gelu_grad_ref(pre)).abs().max()) < 0.5 * 1.26 / 255 + 1e-5


@pytest.mark.parametrize("T", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(200, 192, 128), (64, 64, 72), (37 * 8, 384, 72 * 8), (12544, 1024, 1024)])
def test_linear_fwd_epilogues(T, M, N, K):
    x = gen((M, K), 1, 1.0, T)
    w = gen((N, K), 2, 0.05, T)     # asymmetric, non-square: catches transposes
    b = gen((N,), 3)
    ref = x.float() @ w.float().t() + b
    y = ops.linear_fwd(x, w, b, EPI_BIAS)
    assert y.dtype == T
    assert relerr(y.float(), ref) < TOL[T]
    y32 = ops.linear_fwd(x, w, b, EPI_BIAS_F32)
    assert relerr(y32, ref) < (2e-5 if T == torch.float32 else 1e-2)
    act, aux = ops.linear_gelu(x, w, b)
    check_gelu_pair(x, w, b, act, aux, TOL[T])
    resid = gen((M, N), 4)
    rps = 8
    rowscale = gen(((M + rps - 1) // rps,), 5).abs() + 0.5
    out = ops.linear_fwd(x, w, b, EPI_BIAS_RESID, resid=resid, rowscale=rowscale, rows_per_sample=rps)
    rs = rowscale.repeat_interleave(rps)[:M, None]
    assert relerr(out, resid + rs * ref) < (2e-5 if T == torch.float32 else 1e-2)


@pytest.mark.parametrize("T", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(200, 192, 128), (96, 64, 72), (12544, 1024, 4096)])
def test_linear_backward(T, M, N, K):
    dy = gen((M, N), 1, 1.0, T)
    w = gen((N, K), 2, 0.05, T)
    x = gen((M, K), 3, 1.0, T)
    pre = gen((M, K), 4, 1.0, T)
    dx_ref = dy.float() @ w.float()
    dx = ops.linear_dgrad(dy, w)
    assert relerr(dx.float(), dx_ref) < TOL[T]
    dxg = ops.linear_dgrad(dy, w, gelu_aux=gelu_aux_of(pre))
    assert relerr(dxg.float(), dx_ref * gelu_grad_ref(pre)) < TOL[T]
    dw = ops.linear_wgrad(dy, x)
    dw_ref = dy.double().t() @ x.double()
    assert dw.dtype == torch.float32
    assert relerr(dw, dw_ref) < (2e-5 if T == torch.float32 else 1e-4)     # inputs are exact in T; fp32 accumulate
    db = ops.colsum(dy)
    assert relerr(db, dy.double().sum(0)) < 1e-5


@pytest.mark.parametrize("M,N,K", [(328, 264, 256), (256, 256, 128), (1568, 1024, 1024), (520, 776, 384), (3136, 3072, 1024)])
def test_gemm256_fast_path(M, N, K):
    """Shapes that take the 256x256 LDS-DMA bf16 kernel (csrc/gemm256.h): ragged M/N (clamped rows, masked stores), every
    epilogue, K-major x K-major (forward), K-major x M-major (dgrad), M-major x M-major (wgrad, split over M)."""
    T = torch.bfloat16
    x = gen((M, K), 1, 1.0, T)
    w = gen((N, K), 2, 0.05, T)
    b = gen((N,), 3)
    ref = x.float() @ w.float().t() + b
    assert relerr(ops.linear_fwd(x, w, b, EPI_BIAS).float(), ref) < 1e-2
    assert relerr(ops.linear_fwd(x, w, b, EPI_BIAS_F32), ref) < 2e-5 * math.sqrt(K)
    act, aux = ops.linear_gelu(x, w, b)
    check_gelu_pair(x, w, b, act, aux, 1e-2)
    resid = gen((M, N), 4)
    rowscale = gen(((M + 7) // 8,), 5).abs() + 0.5
    out = ops.linear_fwd(x, w, b, EPI_BIAS_RESID, resid=resid, rowscale=rowscale, rows_per_sample=8)
    assert relerr(out, resid + rowscale.repeat_interleave(8)[:M, None] * ref) < 2e-5 * math.sqrt(K)
    # backward of a Linear(K2 -> N2) with contraction lengths that qualify: dgrad contracts over N2, wgrad over M2
    M2, N2, K2 = (M // 128) * 128 if M >= 128 else 128, K, N
    dy = gen((M2, N2), 6, 1.0, T)
    w2 = gen((N2, K2), 7, 0.05, T)
    x2 = gen((M2, K2), 8, 1.0, T)
    pre2 = gen((M2, K2), 9, 1.0, T)
    dx_ref = dy.float() @ w2.float()
    assert relerr(ops.linear_dgrad(dy, w2).float(), dx_ref) < 1e-2
    aux2 = gelu_aux_of(pre2)
    assert relerr(ops.linear_dgrad(dy, w2, gelu_aux=aux2).float(), dx_ref * gelu_aux_values(aux2)) < 1e-2
    dw = ops.linear_wgrad(dy, x2)
    assert relerr(dw, dy.double().t() @ x2.double()) < 1e-4
    # twice -> bit-identical (no atomics, fixed reduction order)
    assert torch.equal(dw, ops.linear_wgrad(dy, x2))


@pytest.mark.parametrize("M,N,K", [(12544, 1024, 1024), (1568, 3072, 1024), (500, 264, 256), (448, 4096, 384), (3136, 1024, 4096)])
def test_gemm256_224_row_tile_bit_identical_to_256_row_tile(M, N, K):
    """The 224 x 256 tile of gemm256 (round 4: 12544 = 56 x 224 rows fill the last round of workgroups, the lower wave row owns three
    32-row blocks) against the 256 x 256 tile on the same operands -- every epilogue of the forward and data-gradient GEMMs, whole and
    ragged M (clamped rows, masked stores, a last 224-row tile that is mostly padding): the K order per output element is the same,
    so the results must be bit-identical.  pa_debug_set(4, 1 / 2) pins the tile height."""
    from painter_amd._lib import lib
    T = torch.bfloat16
    x, w, b = gen((M, K), 1, 1.0, T), gen((N, K), 2, 0.05, T), gen((N,), 3)
    resid = gen((M, N), 4)
    rowscale = gen(((M + 7) // 8,), 5).abs() + 0.5
    dy, w2, aux2 = gen((M, N), 6, 1.0, T), gen((N, K), 7, 0.05, T), gelu_aux_of(gen((M, K), 9, 1.0, T))

    def run():
        act, aux = ops.linear_gelu(x, w, b)
        return (ops.linear_fwd(x, w, b, EPI_BIAS), ops.linear_fwd(x, w, b, EPI_BIAS_F32), act, aux,
                ops.linear_fwd(x, w, b, EPI_BIAS_RESID, resid=resid, rowscale=rowscale, rows_per_sample=8),
                ops.linear_dgrad(dy, w2), ops.linear_dgrad(dy, w2, gelu_aux=aux2))
    try:
        assert lib.pa_debug_set(4, 1) == 0
        ref = run()
        assert lib.pa_debug_set(4, 2) == 0
        got = run()
    finally:
        lib.pa_debug_set(4, 0)
    for a, r in zip(got, ref):
        assert torch.equal(a, r)
    assert relerr(ref[1], x.float() @ w.float().t() + b) < 2e-5 * math.sqrt(K)


@pytest.mark.parametrize("M,N,K", [(12544, 4096, 1024), (12544, 3072, 1024), (12500, 4096, 256), (12544, 16384, 256), (9000, 4096, 128)])
def test_gemm256_mixed_full_and_half_tiles_bit_identical_to_uniform_tiles(M, N, K):
    """Round 6: multi-round launches run whole rounds of 256-row tiles followed by HALF tiles of 128 rows for the remaining rows (gemm256.h,
    MIXED: fc1 forward / fc2 data gradient 48 x 16 full + 2 x 16 half tiles, qkv forward 42 x 12 + 14 x 12, the decoder embedding 48 x 64 +
    2 x 64; a ragged M whose last half tile is mostly padding; a shape whose rule picks a mixed plan with many half tiles).  In a half tile
    both wave rows work on the 128 rows of wave row 0 (row 0: phases 0 / 1, row 1: phases 2 / 3) -- same ascending K order per output element,
    so every epilogue must return the bits of the uniform tiling (pa_debug_set(12, 1) = the round-5 plans), including the column sums the
    fc2 data-gradient epilogue emits (one partial row per (row tile, wave row): another row count, same sums up to fp32 summation order)."""
    from painter_amd._lib import lib
    T = torch.bfloat16
    x, w, b = gen((M, K), 1, 1.0, T), gen((N, K), 2, 0.05, T), gen((N,), 3)
    resid = gen((M, N), 4) if N <= 4096 else None
    rowscale = gen(((M + 7) // 8,), 5).abs() + 0.5
    # data gradient of a Linear(N -> K): dX [M, N] = dY [M, K] . W [K, N] -- the OUTPUT is the wide side, as in fc2's backward
    dy, w2, aux2 = gen((M, K), 6, 1.0, T), gen((K, N), 7, 0.05, T), gelu_aux_of(gen((M, N), 9, 1.0, T))
    pix = N == 16384

    def run():
        act, aux = ops.linear_gelu(x, w, b)
        cs = torch.full((N,), float("nan"), device=DEV)
        res = [ops.linear_fwd(x, w, b, EPI_BIAS), act, aux, ops.linear_dgrad(dy, w2), ops.linear_dgrad(dy, w2, gelu_aux=aux2, colsum_out=cs)]
        if resid is not None:
            res += [ops.linear_fwd(x, w, b, EPI_BIAS_F32), ops.linear_fwd(x, w, b, EPI_BIAS_RESID, resid=resid, rowscale=rowscale, rows_per_sample=8)]
        if pix and M == 12544:
            res.append(ops.linear_pixshuf(x, w, b, 8, 56, 28, 16, 64))
        return res, cs
    try:
        assert lib.pa_debug_set(12, 1) == 0
        ref, cs_ref = run()
        assert lib.pa_debug_set(12, 0) == 0
        got, cs_got = run()
    finally:
        lib.pa_debug_set(12, 0)
    for a, r in zip(got, ref):
        assert torch.equal(a, r)
    assert relerr(cs_got, cs_ref) < 1e-5 and relerr(cs_got, got[4].double().sum(0)) < 4e-3
    assert relerr(ref[0].float(), x.float() @ w.float().t() + b) < 1e-2


@pytest.mark.parametrize("M,N,K", [(12544, 1024, 4096), (1568, 1024, 1024), (500, 256, 264), (96, 128, 136)])
@pytest.mark.parametrize("T", [torch.bfloat16, torch.float32])
def test_linear_dgrad_column_sums_from_the_epilogue(T, M, N, K):
    """pa_linear_dgrad(dx_colsum=...): the fc2 data-gradient GEMM dpre = (dY . W) * gelu'(pre) also returns the column sums of the dpre it
    stores (= fc1's bias gradient; used to be a separate pa_colsum pass over [R, 4D]).  bf16 big shapes take the gemm256 epilogue
    (both tile heights), the rest the generic engine + pa_colsum.  dX must be the bits of the call without the extra output; the sums
    must equal a column sum of dX as stored (fp32 summation order aside).  Without a GELU side input (that combination
    used to leave the sums unwritten on the bf16 fast path) the sums come from a pa_colsum pass over the stored dX."""
    dy, w, pre = gen((M, N), 6, 1.0, T), gen((N, K), 7, 0.05, T), gen((M, K), 9, 1.0, T)
    aux = gelu_aux_of(pre)
    gp = gelu_aux_values(aux)                                                   # the derivative values the kernel multiplies with
    plain = ops.linear_dgrad(dy, w)
    csp = torch.full((K,), float("nan"), device=DEV)
    assert torch.equal(ops.linear_dgrad(dy, w, colsum_out=csp), plain)
    assert relerr(csp, plain.double().sum(0)) < (1e-5 if T == torch.float32 else 4e-3)
    ref = ops.linear_dgrad(dy, w, gelu_aux=aux)
    from painter_amd._lib import lib
    try:
        for knob in (1, 2, 0):
            lib.pa_debug_set(4, knob)
            cs = torch.full((K,), float("nan"), device=DEV)
            dx = ops.linear_dgrad(dy, w, gelu_aux=aux, colsum_out=cs)
            assert torch.equal(dx, ref)
            want = dx.double().sum(0)                         # dx is the ROUNDED copy of what the epilogue summed: bf16 rounding noise, averaged over M rows
            tol = 1e-5 if T == torch.float32 else 4e-3
            assert relerr(cs, want) < tol, (knob, relerr(cs, want))
            assert relerr(cs, ops.colsum(dx)) < tol
            assert relerr(cs, ((dy.double() @ w.double()) * gp).sum(0)) < (2e-5 if T == torch.float32 else 2e-4)
            cs2 = torch.empty_like(cs)
            ops.linear_dgrad(dy, w, gelu_aux=aux, colsum_out=cs2)
            assert torch.equal(cs, cs2)                      # fixed reduction order: bit-stable
    finally:
        lib.pa_debug_set(4, 0)


def _rel64(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("M,N,K", [(12544, 4096, 1024), (12544, 1024, 4096), (3136, 3072, 1024)])
def test_gemm256_bf16_tight_gates_against_fp64(M, N, K):
    """The kernels bench.py times (256x256 LDS-DMA bf16 GEMM, every fused epilogue) with operands that are exact in bf16 against
    an fp64 reference, gated at what the arithmetic allows instead of a blanket bf16 tolerance (a 1e-2 gate would pass a dropped
    bias or a mis-scaled DropPath row): fp32 accumulation of exact bf16 products, so
      * fp32 outputs (residual epilogue, weight gradient) carry accumulation error only   -> gate 5e-6 * sqrt(K)
      * bf16 outputs add ONE rounding of the result (half an ulp = 2^-9 of the value)      -> gate 2^-8 = 3.9e-3 of max |ref|
    Measured on MI355X (round 2): fp32 outputs 2e-7 .. 9e-7, bf16 outputs 1.2e-3 .. 1.9e-3."""
    T = torch.bfloat16
    x = gen((M, K), 1, 1.0, T)
    w = gen((N, K), 2, 0.05, T)
    b = gen((N,), 3)
    ref = x.double() @ w.double().t() + b.double()
    f32_gate, bf16_gate = 5e-6 * math.sqrt(K), 2.0 ** -8
    errs = {}
    errs["bias_bf16"] = _rel64(ops.linear_fwd(x, w, b, EPI_BIAS), ref)
    errs["bias_f32"] = _rel64(ops.linear_fwd(x, w, b, EPI_BIAS_F32), ref)
    # act and the saved derivative against the pre-activation bits the GELU epilogue sees (= the bias epilogue's output: same operands, same
    # accumulation order): each is one bf16 rounding + the 1.5e-7 of the erf approximation away from the fp64 function of those bits
    act, aux = ops.linear_gelu(x, w, b)
    pre = ops.linear_fwd(x, w, b, EPI_BIAS)
    errs["gelu_act_bf16"] = _rel64(act, torch.nn.functional.gelu(pre.double()))
    errs["gelu_aux_bf16"] = _rel64(ops.gelu_aux_decode(aux), gelu_grad_ref(pre))      # the 8-bit code: half a step = 2.5e-3 absolute of max |gelu'| = 1.13 -> 2.2e-3
    resid = gen((M, N), 4)
    rowscale = gen(((M + 1567) // 1568,), 5).abs() + 0.5
    out = ops.linear_fwd(x, w, b, EPI_BIAS_RESID, resid=resid, rowscale=rowscale, rows_per_sample=1568)
    errs["resid_f32"] = _rel64(out, resid.double() + rowscale.repeat_interleave(1568)[:M, None].double() * ref)
    dy = gen((M, N), 6, 1.0, T)
    pre2 = gen((M, K), 9, 1.0, T)
    dx_ref = dy.double() @ w.double()
    errs["dgrad_bf16"] = _rel64(ops.linear_dgrad(dy, w), dx_ref)
    aux2 = gelu_aux_of(pre2)                                      # the decoded code is exact in fp32: the product is rounded once
    errs["dgrad_dgelu_bf16"] = _rel64(ops.linear_dgrad(dy, w, gelu_aux=aux2), dx_ref * gelu_aux_values(aux2))
    errs["wgrad_f32"] = _rel64(ops.linear_wgrad(dy, x), dy.double().t() @ x.double())
    print("gemm256 %s measured:" % ((M, N, K),), {k: "%.2e" % v for k, v in errs.items()})
    for k, v in errs.items():
        assert v < (f32_gate if k.endswith("f32") else bf16_gate), (k, errs)


@pytest.mark.parametrize("T", [torch.float32, torch.bfloat16])
def test_linear_strided_views_and_pixshuf(T):
    """Column-slice inputs (tap concat buffer) and the decoder pixel-shuffle epilogue (models_painter.py:424-428)."""
    B, Hp, Wp, P, C, K = 2, 8, 4, 16, 64, 256
    M = B * Hp * Wp
    big = gen((M, 2 * K), 1, 1.0, T)
    x = big[:, K:]
    w = gen((P * P * C, K), 2, 0.05, T)
    b = gen((P * P * C,), 3)
    out = ops.linear_pixshuf(x, w, b, B, Hp, Wp, P, C)
    ref = (x.float() @ w.float().t() + b).reshape(B, Hp, Wp, P, P, C)
    ref = torch.einsum("nhwpqc->nchpwq", ref).reshape(B, C, Hp * P, Wp * P).permute(0, 2, 3, 1)
    assert relerr(out.float(), ref) < TOL[T]


@pytest.mark.parametrize("T", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("R,D,variant", [(64, 128, 0), (1000, 1024, 0), (1003, 1024, 0), (25088, 1024, 0), (33, 1280, 0), (4099, 1280, 0),
                                         (1000, 1024, 1), (1003, 1024, 1), (25088, 1024, 1), (33, 1280, 1), (4099, 1280, 1)])
def test_layernorm_fwd_bwd(T, R, D, variant):
    """variant (pa_debug_set(10, .)): 0 = the default backward (rows split over the workgroup's waves wherever D >= 1024, batches of 2 rows;
    R = 1003 / 33 / 4099 end in a half-filled batch), 1 = one wave per row everywhere (what D = 128 always runs)."""
    from painter_amd._lib import lib
    saved = lib.pa_debug_get(10)
    lib.pa_debug_set(10, variant)
    try:
        _layernorm_fwd_bwd(T, R, D)
    finally:
        lib.pa_debug_set(10, saved)


def _layernorm_fwd_bwd(T, R, D):
    x = gen((R, D), 1, 2.0) + 0.3
    gamma = 1 + gen((D,), 2, 0.1)
    beta = gen((D,), 3, 0.1)
    xr = x.double().clone().requires_grad_(True)
    gr, br = gamma.double().clone().requires_grad_(True), beta.double().clone().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(xr, (D,), gr, br, 1e-6)
    y, mean, rstd = ops.layernorm_fwd(x, gamma, beta, 1e-6, T)
    assert relerr(y.float(), yr.detach()) < (1e-5 if T == torch.float32 else 1e-2)
    assert relerr(mean, x.double().mean(1)) < 1e-5
    dy = gen((R, D), 4, 1.0, T)
    yr.backward(dy.double())
    dres = gen((R, D), 5)
    rps = 8
    rowscale = gen(((R + rps - 1) // rps,), 6).abs() + 0.5
    dxT = torch.empty((R, D), dtype=T, device=DEV)
    cs = torch.empty((D,), device=DEV)
    dx, gb = ops.layernorm_bwd(dy, x, mean, rstd, gamma, dres=dres, dxT=dxT, rowscale=rowscale, rows_per_sample=rps, dxT_colsum=cs)
    # the fused bias gradient: column sums of rowscale * dx in fp32 (what dxT is rounded from; a pa_colsum pass over dxT sees the rounded values)
    want_cs = ((dres.double() + xr.grad) * rowscale.repeat_interleave(rps)[:R, None].double()).sum(0)
    assert relerr(cs, want_cs) < 2e-5 and relerr(cs, ops.colsum(dxT)) < (1e-5 if T == torch.float32 else 4e-3)
    dx_plain, gb_plain = ops.layernorm_bwd(dy, x, mean, rstd, gamma, dres=dres, dxT=torch.empty_like(dxT), rowscale=rowscale, rows_per_sample=rps)
    assert torch.equal(dx, dx_plain) and torch.equal(gb, gb_plain)            # the extra output changes nothing else
    ref = dres.double() + xr.grad
    assert relerr(dx, ref) < 2e-5
    assert relerr(dxT.float(), ref * rowscale.repeat_interleave(rps)[:R, None].double()) < TOL[T]
    assert relerr(gb[0], gr.grad) < 1e-4
    assert relerr(gb[1], br.grad) < 1e-4
    # in-place accumulate form (dres aliases dx), no T copy
    dx2 = dres.clone()
    ops.layernorm_bwd(dy, x, mean, rstd, gamma, dres=dx2, dx=dx2)
    assert relerr(dx2, ref) < 2e-5


def attn_reference(qkv, rel_h, rel_w, B, L, H, Hp, Wp, scale):
    """Painter/models_painter.py:76-86 + util/vitdet_utils.py:96-125 in fp64 (head dim from the operand shapes)."""
    hd = rel_h.shape[1]
    D = H * hd
    q, k, v = qkv.double().reshape(B, L, 3, H, hd).permute(2, 0, 3, 1, 4).reshape(3, B * H, L, hd).unbind(0)
    attn = (q * scale) @ k.transpose(-2, -1)
    ih = (torch.arange(Hp)[:, None] - torch.arange(Hp)[None, :] + Hp - 1).to(qkv.device)
    iw = (torch.arange(Wp)[:, None] - torch.arange(Wp)[None, :] + Wp - 1).to(qkv.device)
    Rh, Rw = rel_h.double()[ih], rel_w.double()[iw]
    rq = q.reshape(B * H, Hp, Wp, hd)
    bh = torch.einsum("bhwc,hkc->bhwk", rq, Rh)
    bw = torch.einsum("bhwc,wkc->bhwk", rq, Rw)
    attn = (attn.view(B * H, Hp, Wp, Hp, Wp) + bh[..., :, None] + bw[..., None, :]).view(B * H, L, L)
    lse = torch.logsumexp(attn, dim=-1)
    o = attn.softmax(-1) @ v
    return o.view(B, H, L, hd).permute(0, 2, 1, 3).reshape(B * L, D), lse


@pytest.fixture
def attn_generation():
    """pa_attn_set_generation for the duration of one test (0 = default: generation 3 where it applies; 2 = never generation 3)."""
    from painter_amd._lib import lib

    def set_(g):
        assert lib.pa_attn_set_generation(g) == 0
    yield set_
    lib.pa_attn_set_generation(0)


def _attn_cases(shapes):
    """(T, B, H, Hp, Wp, generation): every shape in both builds with the default kernels, plus generation 2 where the switch matters
    (bf16 on a 28-token-wide grid, where generation 3 is the default)."""
    out = []
    for T in (torch.float32, torch.bfloat16):
        for (B, H, Hp, Wp) in shapes:
            out.append((T, B, H, Hp, Wp, 0))
            if T == torch.bfloat16 and Wp == 28:
                out.append((T, B, H, Hp, Wp, 2))
    return out


@pytest.mark.parametrize("T", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("nblocks,Hp,Wp,hd", [(24, 56, 28, 64), (3, 8, 4, 64), (5, 64, 32, 80), (1, 24, 12, 80)])
def test_relpos_pack_batch_equals_the_per_block_packs(T, nblocks, Hp, Wp, hd):
    """pa_relpos_pack_batch (one launch for every block's Rcat and Rcat^T, what a training step runs after the optimizer wrote the tables)
    against pa_relpos_pack / pa_relpos_pack_t block by block: bit-identical, padding rows included; re-packing into the same buffers after
    the tables changed sees the new values."""
    hs = [gen((2 * Hp - 1, hd), 10 + k, 0.2) for k in range(nblocks)]
    ws = [gen((2 * Wp - 1, hd), 50 + k, 0.2) for k in range(nblocks)]
    tabs = torch.tensor([t.data_ptr() for t in hs] + [t.data_ptr() for t in ws], dtype=torch.int64).cuda()
    rcat, rcatT = ops.relpos_pack_batch(tabs, nblocks, Hp, Wp, hd, T)
    nrp = rcat.shape[1]
    assert rcat.shape == (nblocks, nrp, hd) and rcatT.shape == (nblocks, hd, nrp) and nrp % 32 == 0 and nrp >= 2 * Hp + 2 * Wp - 2
    for k in range(nblocks):
        assert torch.equal(rcat[k], ops.relpos_pack(hs[k], ws[k], Hp, Wp, T)), k
        assert torch.equal(rcatT[k], ops.relpos_pack_t(hs[k], ws[k], Hp, Wp, T)), k
    hs[nblocks - 1].mul_(-2.0)
    ws[0].add_(1.0)
    ops.relpos_pack_batch(tabs, nblocks, Hp, Wp, hd, T, rcat=rcat, rcatT=rcatT)
    for k in (0, nblocks - 1):
        assert torch.equal(rcat[k], ops.relpos_pack(hs[k], ws[k], Hp, Wp, T)) and torch.equal(rcatT[k], ops.relpos_pack_t(hs[k], ws[k], Hp, Wp, T))


@pytest.mark.parametrize("T,B,H,Hp,Wp,gen_", _attn_cases([(1, 2, 8, 4), (2, 2, 56, 28), (1, 1, 16, 8), (1, 2, 8, 12), (2, 1, 8, 20), (1, 2, 16, 16),
                                                          (1, 1, 8, 24), (1, 3, 16, 28), (1, 1, 8, 28)]))
def test_attn_fwd(T, B, H, Hp, Wp, gen_, attn_generation):
    attn_generation(gen_)
    L = Hp * Wp
    qkv = gen((B * L, 3 * H * 64), 1, 1.0, T)
    rel_h = gen((2 * Hp - 1, 64), 2, 0.2)
    rel_w = gen((2 * Wp - 1, 64), 3, 0.2)
    rcat = ops.relpos_pack(rel_h, rel_w, Hp, Wp, T)
    assert relerr(rcat[: 2 * Hp - 1].float(), rel_h) < (1e-7 if T == torch.float32 else 1e-2)
    out, lse = ops.attn_fwd(qkv, rcat, B, L, H, Hp, Wp, 0.125)
    ref, lse_ref = attn_reference(qkv, rcat[: 2 * Hp - 1], rcat[2 * Hp - 1: 2 * Hp + 2 * Wp - 2], B, L, H, Hp, Wp, 0.125)
    e_o, e_l = relerr(out.float(), ref), relerr(lse, lse_ref)
    assert e_l < (1e-5 if T == torch.float32 else 2e-3), (e_o, e_l)
    assert e_o < (2e-5 if T == torch.float32 else 1.2e-2), (e_o, e_l)      # bf16: measured <= 7.8e-3 (tools/attn3_diag.py), the output's own rounding is 3.9e-3


@pytest.mark.parametrize("T,B,H,Hp,Wp,gen_", _attn_cases([(1, 2, 8, 4), (2, 2, 56, 28), (3, 1, 16, 8), (1, 2, 8, 12), (2, 1, 8, 20), (1, 2, 16, 16),
                                                          (1, 1, 8, 24), (1, 3, 16, 28), (1, 1, 8, 28)]))
def test_attn_bwd(T, B, H, Hp, Wp, gen_, attn_generation):
    attn_generation(gen_)
    L = Hp * Wp
    nh, nw = 2 * Hp - 1, 2 * Wp - 1
    qkv = gen((B * L, 3 * H * 64), 1, 1.0, T)
    rel_h = gen((nh, 64), 2, 0.2)
    rel_w = gen((nw, 64), 3, 0.2)
    dout = gen((B * L, H * 64), 4, 1.0, T)
    rcat = ops.relpos_pack(rel_h, rel_w, Hp, Wp, T)
    rcatT = ops.relpos_pack_t(rel_h, rel_w, Hp, Wp, T)
    assert torch.equal(rcatT.t().contiguous(), rcat)
    out, lse, tables = ops.attn_fwd(qkv, rcat, B, L, H, Hp, Wp, 0.125, need_tables=True)
    assert (tables is not None) == (gen_ != 2 and T == torch.bfloat16 and Wp == 28 and Hp % 8 == 0)
    dqkv, drcat = ops.attn_bwd(qkv, rcat, rcatT, out, dout, lse, B, L, H, Hp, Wp, 0.125, tables=tables)
    q64 = qkv.double().clone().requires_grad_(True)
    rh64 = rcat[:nh].double().clone().requires_grad_(True)
    rw64 = rcat[nh:nh + nw].double().clone().requires_grad_(True)
    ref, _ = attn_reference(q64, rh64, rw64, B, L, H, Hp, Wp, 0.125)
    ref.backward(dout.double())
    D = H * 64
    tol = 5e-5 if T == torch.float32 else 1.6e-2       # bf16: measured <= 1.05e-2 of the largest entry (tools/attn3_diag.py)
    errs = dict(dq=relerr(dqkv[:, :D].float(), q64.grad[:, :D]), dk=relerr(dqkv[:, D:2 * D].float(), q64.grad[:, D:2 * D]),
                dv=relerr(dqkv[:, 2 * D:].float(), q64.grad[:, 2 * D:]), drh=relerr(drcat[:nh], rh64.grad),
                drw=relerr(drcat[nh:nh + nw], rw64.grad))
    assert max(errs.values()) < tol, errs
    assert float(drcat[nh + nw:].abs().max()) == 0.0 if drcat.shape[0] > nh + nw else True


@pytest.mark.parametrize("T", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,Hp,Wp", [(2, 2, 8, 4), (1, 3, 16, 8), (1, 2, 64, 32), (2, 1, 8, 12), (1, 2, 8, 32), (2, 1, 16, 28), (1, 1, 4, 32)])
def test_attn_head_dim_80_fwd_bwd(T, B, H, Hp, Wp):
    """head_dim 80 (ViT-H/14, BASELINE configs[4]: 1280 / 16 heads; its token grid is 64 x 32): five 16-deep k-steps in Q.K^T / dO.V^T,
    three 32-row output blocks of which the last is half padding -- forward (out, lse) and backward (dq, dk, dv, d rel_pos_h,
    d rel_pos_w) against the fp64 reference of models_painter.py:76-86 + vitdet_utils.py:96-125.  Gates as for head_dim 64.
    Kernels: bf16 with key rows of 12..28 or exactly 32 tokens -> the generation-2 kernels instantiated for HD = 80 (round 4:
    csrc/attn2.hip over attn_tile_hd.h; key rows of 32 = one tile per key row, WP32); everything else (fp32; rows of 4 / 8 tokens) -> the
    generic kernels of attn_fwd.hip / attn_bwd.hip."""
    hd = 80
    L = Hp * Wp
    nh, nw = 2 * Hp - 1, 2 * Wp - 1
    scale = hd ** -0.5
    qkv = gen((B * L, 3 * H * hd), 1, 1.0, T)
    rel_h, rel_w = gen((nh, hd), 2, 0.2), gen((nw, hd), 3, 0.2)
    dout = gen((B * L, H * hd), 4, 1.0, T)
    rcat = ops.relpos_pack(rel_h, rel_w, Hp, Wp, T)
    rcatT = ops.relpos_pack_t(rel_h, rel_w, Hp, Wp, T)
    assert rcat.shape[1] == hd and torch.equal(rcatT.t().contiguous(), rcat)
    out, lse, tables = ops.attn_fwd(qkv, rcat, B, L, H, Hp, Wp, scale, need_tables=True)
    assert tables is None and out.shape == (B * L, H * hd)
    q64 = qkv.double().clone().requires_grad_(True)
    rh64 = rcat[:nh].double().clone().requires_grad_(True)
    rw64 = rcat[nh:nh + nw].double().clone().requires_grad_(True)
    ref, lse_ref = attn_reference(q64, rh64, rw64, B, L, H, Hp, Wp, scale)
    e_o, e_l = relerr(out.float(), ref.detach()), relerr(lse, lse_ref.detach())
    assert e_l < (1e-5 if T == torch.float32 else 2e-3), (e_o, e_l)
    assert e_o < (2e-5 if T == torch.float32 else 1.2e-2), (e_o, e_l)
    dqkv, drcat = ops.attn_bwd(qkv, rcat, rcatT, out, dout, lse, B, L, H, Hp, Wp, scale)
    ref.backward(dout.double())
    D = H * hd
    tol = 5e-5 if T == torch.float32 else 1.6e-2
    errs = dict(dq=relerr(dqkv[:, :D].float(), q64.grad[:, :D]), dk=relerr(dqkv[:, D:2 * D].float(), q64.grad[:, D:2 * D]),
                dv=relerr(dqkv[:, 2 * D:].float(), q64.grad[:, 2 * D:]), drh=relerr(drcat[:nh], rh64.grad),
                drw=relerr(drcat[nh:nh + nw], rw64.grad))
    assert max(errs.values()) < tol, errs
    assert drcat.shape == (rcat.shape[0], hd) and (drcat.shape[0] == nh + nw or float(drcat[nh + nw:].abs().max()) == 0.0)
    # same input twice -> the same bits (no atomics on a float accumulation order that varies)
    out2, lse2 = ops.attn_fwd(qkv, rcat, B, L, H, Hp, Wp, scale)
    assert torch.equal(out, out2) and torch.equal(lse, lse2)


def test_attn_fwd_spiked_key_online_softmax():
    """Force large running-max jumps late in the key stream (rule: a data-dependent rescale needs its own test)."""
    B, H, Hp, Wp = 1, 1, 16, 8
    L = Hp * Wp
    qkv = gen((B * L, 3 * 64), 7, 0.5)
    qkv[5, 0:64] = 3.0            # q row 5
    qkv[100, 64:128] = 3.0        # k row 100 -> logit spike 0.125*9*64 = 72 at tile 3
    rel = torch.zeros((2 * Hp - 1, 64), device=DEV), torch.zeros((2 * Wp - 1, 64), device=DEV)
    rcat = ops.relpos_pack(rel[0], rel[1], Hp, Wp, torch.float32)
    out, lse = ops.attn_fwd(qkv, rcat, B, L, H, Hp, Wp, 0.125)
    ref, lse_ref = attn_reference(qkv, rel[0], rel[1], B, L, H, Hp, Wp, 0.125)
    assert relerr(out, ref) < 2e-5 and relerr(lse, lse_ref) < 1e-5


def test_attn2_bf16_spiked_key_rebase():
    """bf16 generation-2 forward: the running max is re-based only when a tile exceeds it by > 2^6; force that late."""
    B, H, Hp, Wp = 1, 1, 8, 16
    L = Hp * Wp
    qkv = gen((B * L, 3 * 64), 7, 0.5)
    qkv[5, 0:64] = 3.0
    qkv[100, 64:128] = 3.0        # logit spike 0.125 * 9 * 64 = 72 in the last key tile
    qkv[37, 0:64] = -2.0
    qkv[70, 64:128] = -2.0        # and a smaller one (32) in tile 2
    qkv = qkv.to(torch.bfloat16)
    rel_h, rel_w = gen((2 * Hp - 1, 64), 2, 0.2), gen((2 * Wp - 1, 64), 3, 0.2)
    rcat = ops.relpos_pack(rel_h, rel_w, Hp, Wp, torch.bfloat16)
    out, lse = ops.attn_fwd(qkv, rcat, B, L, H, Hp, Wp, 0.125)
    ref, lse_ref = attn_reference(qkv, rcat[: 2 * Hp - 1], rcat[2 * Hp - 1: 2 * Hp + 2 * Wp - 2], B, L, H, Hp, Wp, 0.125)
    assert relerr(out.float(), ref) < 2e-2 and relerr(lse, lse_ref) < 2e-3


def _attn3_inputs(B, H, Hp, Wp, spike=False):
    L = Hp * Wp
    qkv = gen((B * L, 3 * H * 64), 11, 1.0)
    if spike:                          # key L-5 answers query 7 with a logit far above everything seen before (last key tile)
        qkv[7, 0:64] = 3.0
        qkv[L - 5, H * 64: H * 64 + 64] = 3.0
        qkv[40, 0:64] = -2.0
        qkv[3 * 28 + 9, H * 64: H * 64 + 64] = -2.0
    qkv = qkv.to(torch.bfloat16)
    rel_h, rel_w = gen((2 * Hp - 1, 64), 2, 0.2), gen((2 * Wp - 1, 64), 3, 0.2)
    rcat = ops.relpos_pack(rel_h, rel_w, Hp, Wp, torch.bfloat16)
    rcatT = ops.relpos_pack_t(rel_h, rel_w, Hp, Wp, torch.bfloat16)
    dout = gen((B * L, H * 64), 4, 1.0, torch.bfloat16)
    return L, qkv, rcat, rcatT, dout


@pytest.mark.parametrize("gen_", [0])
def test_attn3_bf16_spiked_key_rebase(gen_, attn_generation):
    """generation-3 forward + backward with a late, large logit (forces the running-max re-base) vs the fp64 reference."""
    attn_generation(gen_)
    B, H, Hp, Wp = 1, 1, 8, 28
    L, qkv, rcat, rcatT, dout = _attn3_inputs(B, H, Hp, Wp, spike=True)
    out, lse, tables = ops.attn_fwd(qkv, rcat, B, L, H, Hp, Wp, 0.125, need_tables=True)
    assert tables is not None
    ref, lse_ref = attn_reference(qkv, rcat[: 2 * Hp - 1], rcat[2 * Hp - 1: 2 * Hp + 2 * Wp - 2], B, L, H, Hp, Wp, 0.125)
    assert relerr(out.float(), ref) < 2e-2 and relerr(lse, lse_ref) < 2e-3
    dqkv, drcat = ops.attn_bwd(qkv, rcat, rcatT, out, dout, lse, B, L, H, Hp, Wp, 0.125, tables=tables)
    q64 = qkv.double().clone().requires_grad_(True)
    o64, _ = attn_reference(q64, rcat[: 2 * Hp - 1].double(), rcat[2 * Hp - 1: 2 * Hp + 2 * Wp - 2].double(), B, L, H, Hp, Wp, 0.125)
    o64.backward(dout.double())
    assert relerr(dqkv.float(), q64.grad) < 3e-2


def test_attn_generations_agree(attn_generation):
    """The two bf16 generations on identical inputs (ViT-L grid, B' = 2): they share no bias code path (k-space tables + VALU adds
    vs one-hot contraction on the matrix pipe), so a dropped or mis-indexed rel-pos term in either shows up here at full size.
    The bound is a few bf16 roundings of the bias tables (generation 3 rounds the kw table to bf16, generation 2 keeps it fp32)."""
    B, H, Hp, Wp = 2, 2, 56, 28
    L, qkv, rcat, rcatT, dout = _attn3_inputs(B, H, Hp, Wp)
    res = {}
    for g_ in (2, 3, 0):
        attn_generation(g_)
        out, lse, tables = ops.attn_fwd(qkv, rcat, B, L, H, Hp, Wp, 0.125, need_tables=True)
        dqkv, drcat = ops.attn_bwd(qkv, rcat, rcatT, out, dout, lse, B, L, H, Hp, Wp, 0.125, tables=tables)
        res[g_] = (out.float(), lse, dqkv.float(), drcat)
    for g3 in (0, 3):
        assert relerr(res[g3][1], res[2][1]) < 4e-3      # each is ~1.3e-3 from the fp64 reference (bf16 bias tables), in different directions
        for a, b in zip(res[g3], res[2]):
            assert relerr(a, b) < 1.5e-2, [relerr(x, y) for x, y in zip(res[g3], res[2])]
    # 0 (default) and 3 (explicit) select the same generation-3 kernels: bit-identical
    for a, b in zip(res[0], res[3]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("B,H,Hp,Wp", [(2, 2, 56, 28), (1, 3, 16, 28), (1, 1, 8, 28)])
def test_attn3_fused_relpos_gradient_vs_dG_gemm(B, H, Hp, Wp):
    """The rel-pos table gradient contracted inside the generation-3 dQ kernel (one fp32 partial per workgroup, fixed-order sum; no dG
    in HBM) against the older route -- dG written as bf16, gather GEMM over it -- on the same inputs: both contract the same bf16 dG
    entries with the same bf16 Q, so they agree to fp32 summation order; dQ / dK / dV must be bit-identical (the key loop is shared).
    Both against the fp64 reference too, and the fused route twice for bit-stability."""
    from painter_amd._lib import PA_BF16, lib
    L, qkv, rcat, rcatT, dout = _attn3_inputs(B, H, Hp, Wp)
    nh, nw = 2 * Hp - 1, 2 * Wp - 1
    out, lse, tables = ops.attn_fwd(qkv, rcat, B, L, H, Hp, Wp, 0.125, need_tables=True)
    res = {}
    try:
        for mode in (2, 1, 2):
            assert lib.pa_debug_set(7, mode) == 0
            nb = lib.pa_attn_bwd_relpos_partials_bytes(PA_BF16, B, L, H, Hp, Wp, 64)
            assert (nb > 0) == (mode == 2)
            dqkv, dg = ops.attn_bwd_core(qkv, rcat, rcatT, out, dout, lse, B, L, H, Hp, Wp, 0.125, tables=tables)
            assert (dg.dtype == torch.uint8) == (mode == 2)
            drcat = ops.attn_bwd_relpos(dg, qkv, rcat.shape[0], B, L, H, Hp, Wp)
            if mode in res:
                assert torch.equal(res[mode][0], dqkv) and torch.equal(res[mode][1], drcat)
            res[mode] = (dqkv.clone(), drcat.clone())
    finally:
        lib.pa_debug_set(7, 0)
    assert torch.equal(res[1][0], res[2][0])
    assert relerr(res[2][1], res[1][1]) < 1e-5, relerr(res[2][1], res[1][1])
    q64 = qkv.double().clone().requires_grad_(True)
    rh64 = rcat[:nh].double().clone().requires_grad_(True)
    rw64 = rcat[nh:nh + nw].double().clone().requires_grad_(True)
    ref, _ = attn_reference(q64, rh64, rw64, B, L, H, Hp, Wp, 0.125)
    ref.backward(dout.double())
    for mode in (1, 2):
        e = (relerr(res[mode][1][:nh], rh64.grad), relerr(res[mode][1][nh:nh + nw], rw64.grad))
        print("rel-pos table gradient vs fp64, mode %d (1 = dG + GEMM, 2 = fused): %.3e %.3e" % ((mode,) + e))
        assert max(e) < 1.6e-2
    assert float(res[2][1][nh + nw:].abs().max()) == 0.0


@pytest.mark.parametrize("B,H,Hp,Wp", [(2, 2, 56, 28), (1, 3, 16, 28)])
def test_attn3_delta_inside_the_dq_kernel_vs_the_prep_launch(B, H, Hp, Wp):
    """Round 5: the forward writes the log-sum-exp fields of the table tiles and the dQ kernel computes Delta = rowsum(dO o O) of its own
    rows (and leaves -Delta in the tiles for the dKV launch) -- the separate prep launch per block is gone.  Against the round-4 route
    (pa_attn_bwd_prep in front) on the same inputs: another fp32 summation order of the 64 products per row, so the outputs agree to a
    bf16 ulp here and there, not bit for bit; both are held to the fp64 reference by the tests above.  The fused route twice: bit-stable."""
    L, qkv, rcat, rcatT, dout = _attn3_inputs(B, H, Hp, Wp)
    out, lse, tables = ops.attn_fwd(qkv, rcat, B, L, H, Hp, Wp, 0.125, need_tables=True)
    res = {}
    for mode in ("fused", "launch", "fused"):
        t = tables.clone()                                   # the forward's tiles; each route adds its own Delta field
        dqkv, dg = ops.attn_bwd_core(qkv, rcat, rcatT, out, dout, lse, B, L, H, Hp, Wp, 0.125, tables=t, prep=mode)
        drcat = ops.attn_bwd_relpos(dg, qkv, rcat.shape[0], B, L, H, Hp, Wp)
        if mode in res:
            assert torch.equal(res[mode][0], dqkv) and torch.equal(res[mode][1], drcat)
        res[mode] = (dqkv.clone(), drcat.clone())
    e = (relerr(res["fused"][0].float(), res["launch"][0].float()), relerr(res["fused"][1], res["launch"][1]))
    print("Delta inside dQ vs prep launch: dqkv %.2e, d rel_pos %.2e" % e)
    assert e[0] < 8e-3 and e[1] < 2e-3, e


@pytest.mark.parametrize("gen_", [0])
def test_attn3_deterministic(gen_, attn_generation):
    attn_generation(gen_)
    B, H, Hp, Wp = 1, 2, 16, 28
    L, qkv, rcat, rcatT, dout = _attn3_inputs(B, H, Hp, Wp)
    runs = []
    for _ in range(2):
        out, lse, tables = ops.attn_fwd(qkv, rcat, B, L, H, Hp, Wp, 0.125, need_tables=True)
        dqkv, drcat = ops.attn_bwd(qkv, rcat, rcatT, out, dout, lse, B, L, H, Hp, Wp, 0.125, tables=tables)
        runs.append((out.clone(), lse.clone(), dqkv.clone(), drcat.clone()))
    for a, b in zip(*runs):
        assert torch.equal(a, b)


def test_determinism_same_input_bit_identical():
    x = gen((777, 1024), 1, 1.0, torch.bfloat16)
    w = gen((1024, 1024), 2, 0.05, torch.bfloat16)
    dy = gen((777, 1024), 3, 1.0, torch.bfloat16)
    a = ops.linear_wgrad(dy, x).clone()
    b = ops.linear_wgrad(dy, x).clone()
    assert torch.equal(a, b)
    c, d = ops.linear_dgrad(dy, w).clone(), ops.linear_dgrad(dy, w).clone()
    assert torch.equal(c, d)


@pytest.mark.parametrize("B,Hi,Wi", [(2, 16, 128), (1, 32, 64), (2, 64, 192)])
def test_conv64_bf16_tile_kernels(B, Hi, Wi):
    """Dedicated bf16 3x3 conv kernels (csrc/conv64.h; Wi % 64 == 0 takes them): fused decoder tail forward, data gradient with
    the inverse pixel shuffle, weight gradient -- against torch conv2d on the same bf16-rounded operands
    (models_painter.py:328-333, :430)."""
    import torch.nn.functional as F
    T = torch.bfloat16
    P, Hp, Wp = 16, Hi // 16, Wi // 16
    x = gen((B, Hi, Wi, 64), 1, 1.0, T)                       # NHWC
    w3 = gen((64, 64, 3, 3), 2, 0.05)
    b3, gamma, beta = gen((64,), 3, 0.1), 1.0 + gen((64,), 4, 0.1), gen((64,), 5, 0.1)
    w1, b1 = gen((3, 64), 6, 0.1), gen((3,), 7, 0.1)
    w3r, wf = ops.conv3x3_pack(w3, T)
    pred, y3 = ops.decoder_tail_fwd(x, w3r, b3, gamma, beta, w1, b1, 1e-6, save_y3=True)
    xn = x.float().permute(0, 3, 1, 2)
    w3b = w3.to(T).float()
    y_ref = F.conv2d(xn, w3b, b3, padding=1)
    assert relerr(y3.float().permute(0, 3, 1, 2), y_ref) < 1e-2
    yq = y3.float().permute(0, 3, 1, 2)                         # the kernel normalises the bf16-rounded conv output
    u = yq.mean(1, keepdim=True)
    s = (yq - u).pow(2).mean(1, keepdim=True)
    z = (yq - u) / torch.sqrt(s + 1e-6) * gamma[None, :, None, None] + beta[None, :, None, None]
    pred_ref = F.conv2d(F.gelu(z), w1[:, :, None, None], b1)
    assert relerr(pred, pred_ref) < 2e-3
    # data gradient + inverse pixel shuffle
    dy = gen((B, Hi, Wi, 64), 8, 1.0, T)
    dE = ops.conv3x3_dgrad_unshuffle(dy, wf, B, Hp, Wp, P)
    dx_ref = F.conv_transpose2d(dy.float().permute(0, 3, 1, 2), w3b, padding=1)          # [B,64,Hi,Wi]
    dx_tok = dx_ref.reshape(B, 64, Hp, P, Wp, P).permute(0, 2, 4, 3, 5, 1).reshape(B * Hp * Wp, P * P * 64)
    assert relerr(dE.float(), dx_tok) < 1e-2
    # weight gradient (fp32 accumulate, deterministic)
    dw = ops.conv3x3_wgrad(dy, x)
    dyn = dy.float().permute(0, 3, 1, 2)
    dw_ref = torch.nn.grad.conv2d_weight(xn, (64, 64, 3, 3), dyn, padding=1)
    assert relerr(dw, dw_ref) < 1e-4
    assert torch.equal(dw, ops.conv3x3_wgrad(dy, x))
    # round 5: the weight-gradient workgroups walk DOWN 64-pixel column strips in 2-row steps over a ring of four halo rows in LDS (every
    # input row fetched once, the next step's loads in flight under the MFMAs).  At these sizes every workgroup has one tile; capping the
    # number of workgroups makes each walk many -- both ring phases, fresh starts at strip boundaries in the middle of a run, uneven runs
    from painter_amd._lib import lib
    try:
        for cap in (1, 3, 5):
            assert lib.pa_debug_set(9, cap) == 0
            dwc = ops.conv3x3_wgrad(dy, x)
            assert relerr(dwc, dw_ref) < 1e-4, (cap, relerr(dwc, dw_ref))
            assert relerr(dwc, dw) < 1e-5                                 # the same products, another fp32 summation order across workgroups (measured 3e-6)
    finally:
        lib.pa_debug_set(9, 0)


@pytest.mark.parametrize("T", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,Hi,Wi", [(2, 16, 64), (3, 192, 128)])
def test_decoder_tail_pointwise_backward_vs_autograd(T, B, Hi, Wi):
    """pa_decoder_tail_bwd_pointwise (round 5 rewrite: 8 lanes x 8 channels per pixel, float2 arithmetic, inputs two iterations ahead):
    the backward through Conv1x1(64 -> 3) . GELU . LayerNorm2D(64) (models_painter.py:328-333, util/vitdet_utils.py:204-209) against
    torch autograd in float64 on the same (T-rounded) conv output -- dy3 and the four parameter gradients.  The larger shape makes the
    2048 persistent workgroups take a second, partial, grid-stride iteration that crosses sample boundaries."""
    import torch.nn.functional as F
    y3 = gen((B, Hi, Wi, 64), 1, 1.0, T)
    dpred = gen((B, 3, Hi, Wi), 2, 1.0)
    gamma, beta = 1.0 + gen((64,), 4, 0.1), gen((64,), 5, 0.1)
    w1 = gen((3, 64), 6, 0.1)
    dy3, grads = ops.decoder_tail_bwd_pointwise(dpred, y3, gamma, beta, w1, 1e-6)
    y = y3.double().permute(0, 3, 1, 2).clone().requires_grad_(True)
    g64, b64, w64 = (t.double().clone().requires_grad_(True) for t in (gamma, beta, w1))
    b1 = torch.zeros(3, dtype=torch.float64, device=DEV, requires_grad=True)
    u = y.mean(1, keepdim=True)
    v = (y - u).pow(2).mean(1, keepdim=True)
    z = (y - u) / torch.sqrt(v + 1e-6) * g64[None, :, None, None] + b64[None, :, None, None]
    pred = F.conv2d(F.gelu(z), w64[:, :, None, None], b1)
    pred.backward(dpred.double())
    tol = 2e-5 if T == torch.float32 else 1e-2
    e = {"dy3": relerr(dy3.float().permute(0, 3, 1, 2), y.grad), "dgamma": relerr(grads[0:64], g64.grad), "dbeta": relerr(grads[64:128], b64.grad),
         "dw1": relerr(grads[128:320].reshape(3, 64), w64.grad), "db1": relerr(grads[320:323], b1.grad)}
    print("decoder tail backward %s %s:" % (T, (B, Hi, Wi)), {k: "%.2e" % v_ for k, v_ in e.items()})
    assert e["dy3"] < tol and max(e["dgamma"], e["dbeta"], e["dw1"], e["db1"]) < (2e-5 if T == torch.float32 else 2e-4), e
    dy3b, gradsb = ops.decoder_tail_bwd_pointwise(dpred, y3, gamma, beta, w1, 1e-6)
    assert torch.equal(dy3, dy3b) and torch.equal(grads, gradsb)           # fixed reduction order: bit-stable


def test_layernorm_bwd_bitstable_beside_concurrent_mfma_kernels():
    """Regression for the SLP / packed-fp32 hazard (painter_amd/build.py): LayerNorm backward on the current
    stream, fixed inputs, while MFMA weight-gradient GEMMs run on a second stream -- every launch must be bit-identical to the
    undisturbed one (with SLP-vectorised code ~1 % of the launches returned one wrong row)."""
    T = torch.bfloat16
    R, D = 3136, 1024
    x = gen((R, D), 1)
    gam, bet = 1 + gen((D,), 2, 0.1), gen((D,), 3, 0.1)
    _, mean, rstd = ops.layernorm_fwd(x, gam, bet, 1e-6, T)
    dy = gen((R, 4 * D), 4, 1.0, T)[:, D:2 * D]
    dres0 = gen((R, D), 5)
    sdy, sx = gen((R, 4096), 6, 1.0, T), gen((R, 1024), 7, 1.0, T)          # M = 3136: generic-engine wgrad, as in the model at B=2
    side = torch.cuda.Stream()

    def run(load):
        if load:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    ops.linear_wgrad(sdy, sx)
        outs = []
        for _ in range(6):
            dres = dres0.clone()
            dxT = torch.empty(R, D, dtype=T, device=DEV)
            dx, gb = ops.layernorm_bwd(dy, x, mean, rstd, gam, dres=dres, dx=dres, dxT=dxT)
            outs.append((dx, dxT, gb))
        if load:
            torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        return outs

    ref = run(False)[0]
    bad = 0
    for _ in range(120):
        for o in run(True):
            bad += int(not (torch.equal(o[0], ref[0]) and torch.equal(o[1], ref[1]) and torch.equal(o[2], ref[2])))
    assert bad == 0, "%d of 720 launches differ" % bad


def _beside_mfma_load(fn, ref, n_outer, per_outer):
    """Runs fn() per_outer times per round on the current stream while bf16 MFMA weight-gradient GEMMs (gemm256) run on a second
    stream; -> number of runs whose outputs are not bit-identical to `ref` (counted on the device: no host sync inside the loop)."""
    sdy, sx = gen((12544, 1024), 6, 1.0, torch.bfloat16), gen((12544, 1024), 7, 1.0, torch.bfloat16)
    side = torch.cuda.Stream()
    bad = torch.zeros((), dtype=torch.int64, device=DEV)
    for _ in range(n_outer):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                ops.linear_wgrad(sdy, sx)
        for _ in range(per_outer):
            o = fn()
            diff = torch.zeros((), dtype=torch.bool, device=DEV)
            for a, b in zip(o, ref):
                diff = diff | torch.ne(a.view(torch.uint8), b.view(torch.uint8)).any()
            bad += diff
        torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    return int(bad.item())


@pytest.mark.parametrize("gen_", [0, 2])
def test_attention_bitstable_beside_concurrent_mfma_kernels(gen_, attn_generation):
    """The attention forward / backward kernels beside another stream's MFMA workgroups (the default two-stream backward): 3000+ kernel
    launches, every one bit-identical to the undisturbed result.  (LayerNorm backward once failed this way with packed-fp32 VALU code,
    DESIGN.md section 6; the attention kernels carry no packed fp32 arithmetic any more.)"""
    attn_generation(gen_)
    B, H, Hp, Wp = 1, 4, 56, 28
    L, qkv, rcat, rcatT, dout = _attn3_inputs(B, H, Hp, Wp)

    def once():
        out, lse, tables = ops.attn_fwd(qkv, rcat, B, L, H, Hp, Wp, 0.125, need_tables=True)
        dqkv, dG = ops.attn_bwd_core(qkv, rcat, rcatT, out, dout, lse, B, L, H, Hp, Wp, 0.125, tables=tables)
        return out, lse, dqkv, dG

    ref = once()
    torch.cuda.synchronize()
    bad = _beside_mfma_load(once, ref, 125, 6)               # 750 x (fwd, delta/prep, dq, dkv)
    assert bad == 0, "%d of 750 runs differ" % bad


def test_gemm256_epilogues_bitstable_beside_concurrent_mfma_kernels():
    """Same for the fused GEMM epilogues (bias + erf-GELU, bias + DropPath scale + residual, dGELU): fp32 VALU code of one kernel next to
    another kernel's MFMA workgroups."""
    T = torch.bfloat16
    M, N, K = 3136, 1024, 1024
    x, w, b = gen((M, K), 1, 1.0, T), gen((N, K), 2, 0.05, T), gen((N,), 3)
    resid = gen((M, N), 4)
    rowscale = gen((2,), 5).abs() + 0.5
    aux8 = gelu_aux_of(gen((M, K), 9, 1.0, T))
    dy = gen((M, N), 6, 1.0, T)

    def once():
        act, p_ = ops.linear_gelu(x, w, b)
        out = ops.linear_fwd(x, w, b, EPI_BIAS_RESID, resid=resid, rowscale=rowscale, rows_per_sample=1568)
        dx = ops.linear_dgrad(dy, w, gelu_aux=aux8)           # (any codes serve as the saved derivative here: bit-stability only)
        return act, p_, out, dx

    ref = once()
    torch.cuda.synchronize()
    bad = _beside_mfma_load(once, ref, 125, 8)               # 1000 x 3 GEMM launches
    assert bad == 0, "%d of 1000 runs differ" % bad


@pytest.mark.parametrize("P", [16, 14])
@pytest.mark.parametrize("T", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("seggpt", [False, True])
def test_patch_embed_and_token_assembly_in_isolation(T, seggpt, P):
    """pa_patch_embed_fwd / pa_patch_embed_wgrad alone: Conv2d(3, D, 16, 16) as an im2col GEMM with the token
    assembly in its epilogue (util/vitdet_utils.py:182-186; models_painter.py:387-409: mask token on the masked target patches, segment
    tokens, abs pos; models_seggpt.py:415-420: type tokens) against torch's conv2d in fp64, and the weight gradient against autograd."""
    B, Hp, Wp, D = 2, 4, 6, 128          # P = 14 (ViT-H/14): K = 588 is no multiple of 8 -> the generic gather + the zero-padded weight pack
    L = Hp * Wp
    imgs, tgts = gen((B, 3, Hp * P, Wp * P), 1), gen((B, 3, Hp * P, Wp * P), 2)
    w, bias = gen((D, 3, P, P), 3, 0.05), gen((D,), 4, 0.1)
    mask_token, seg_x, seg_y = gen((D,), 5, 0.3), gen((D,), 6, 0.3), gen((D,), 7, 0.3)
    pos = gen((L, D), 8, 0.2)
    tcls, tins = gen((D,), 9, 0.3), gen((D,), 10, 0.3)
    seg_type = torch.tensor([0.0, 1.0], device=DEV)
    mask = (torch.rand(B, L, generator=torch.Generator().manual_seed(11)) < 0.4).to(DEV)
    wq = w.to(T)
    wop = ops.patch_weight_pack(w, T, P)
    assert wop.shape == (D, (3 * P * P + 7) // 8 * 8) and torch.equal(wop[:, :3 * P * P], wq.reshape(D, -1)) and float(wop[:, 3 * P * P:].float().abs().sum()) == 0.0
    x = ops.patch_embed_fwd(T, imgs, tgts, wop if P % 8 else wq.reshape(D, -1).contiguous(), bias, mask_token, seg_x, seg_y, pos, mask.to(torch.uint8),
                            tcls if seggpt else None, tins if seggpt else None, seg_type if seggpt else None, B, Hp, Wp, P, D)
    conv = lambda im: torch.nn.functional.conv2d(im.to(T).double(), wq.double(), bias.double(), stride=P).permute(0, 2, 3, 1).reshape(B, L, D)
    ex, ey = conv(imgs), conv(tgts)
    m = mask.double()[:, :, None]
    ey = ey * (1 - m) + mask_token.double() * m
    ex, ey = ex + seg_x.double(), ey + seg_y.double()
    ex, ey = ex + pos.double(), ey + pos.double()
    if seggpt:
        st = seg_type.double()[:, None, None]                       # models_seggpt.py:415-420: type 0 -> type_token_cls, type 1 -> type_token_ins
        tt = tcls.double() * (st == 0) + tins.double() * (st == 1)
        ex, ey = ex + tt, ey + tt
    ref = torch.cat([ex, ey], 0).reshape(2 * B * L, D)
    assert relerr(x, ref) < (2e-6 if T == torch.float32 else 3e-3), relerr(x, ref)
    # weight gradient: dW[d, c, i, j] = sum over both streams and patches of dpe * pixel (masked target patches contribute nothing upstream:
    # tokens_bwd zeroes them before this GEMM, here dpe is given)
    dpe = gen((2 * B * L, D), 12, 1.0, T)
    dw = ops.patch_embed_wgrad(dpe, imgs, tgts, B, Hp, Wp, P, D)
    cols = lambda im: torch.nn.functional.unfold(im.to(T).double(), P, stride=P).transpose(1, 2).reshape(B * L, 3 * P * P)
    refw = dpe.double()[:B * L].t() @ cols(imgs) + dpe.double()[B * L:].t() @ cols(tgts)
    assert relerr(dw, refw) < (2e-6 if T == torch.float32 else 2e-5), relerr(dw, refw)
    if ops.patch_cols_ok(T, B, L, P, D):
        # the bf16 fast path of the engine: materialised im2col operand (bit-exact: index math + one rounding) + the 256 x 256 GEMM with
        # the same token-assembly epilogue; the weight gradient is the ordinary nn.Linear weight gradient on that operand
        cm = ops.patch_im2col(imgs, tgts, B, Hp, Wp, P)
        assert torch.equal(cm.float().cpu(), torch.cat([cols(imgs), cols(tgts)]).float().cpu())
        x2 = ops.patch_embed_fwd_cols(cm, wop, bias, mask_token, seg_x, seg_y, pos, mask.to(torch.uint8), tcls if seggpt else None,
                                      tins if seggpt else None, seg_type if seggpt else None, B, L, D)
        assert relerr(x2, ref) < 3e-3, relerr(x2, ref)
        assert relerr(ops.linear_wgrad(dpe, cm), refw) < 2e-5
    else:
        assert T == torch.float32 or P % 8


@pytest.mark.parametrize("src,Hp,Wp", [(14, 56, 28), (16, 64, 32), (14, 8, 4), (14, 14, 14)])
def test_abs_pos_resize_operator_sparse_rows_fwd_bwd(src, Hp, Wp):
    """pa_pos_fwd / pa_pos_bwd (get_abs_pos, util/vitdet_utils.py:128-157, as the constant operator M in row-sparse form) against
    F.interpolate(bicubic, align_corners=False) itself and its autograd, in fp64."""
    from painter_amd import hostmath
    D, L, S = 96, Hp * Wp, src * src
    M = hostmath.abs_pos_operator(src, Hp, Wp)
    dev = lambda t: (torch.from_numpy(t[0]).to(DEV), torch.from_numpy(t[1]).to(DEV))
    fwd, bwd = dev(hostmath.sparse_rows(M)), dev(hostmath.sparse_rows(M.T))
    pe = gen((S, D), 1)
    pos = ops.pos_fwd(fwd, pe, L, D)
    pe64 = pe.double().cpu().clone().requires_grad_(True)
    ref = torch.nn.functional.interpolate(pe64.reshape(1, src, src, D).permute(0, 3, 1, 2), size=(Hp, Wp), mode="bicubic", align_corners=False) \
        if (Hp, Wp) != (src, src) else pe64.reshape(1, src, src, D).permute(0, 3, 1, 2)
    ref = ref.permute(0, 2, 3, 1).reshape(L, D)
    assert relerr(pos, ref.detach()) < 2e-6
    gx, gy = gen((L, D), 2), gen((L, D), 3)
    dpe = torch.empty((S, D), device=DEV)
    ops.pos_bwd(bwd, gx, gy, dpe, S, D)
    ref.backward((gx + gy).double().cpu())
    assert relerr(dpe, pe64.grad) < 2e-6


def test_c_abi_of_the_hot_path_rejects_bad_shapes_instead_of_reading_out_of_bounds():
    """Error behaviour at the boundary: every entry point returns a hipError (the Python binding raises) for a shape its kernels cannot
    take -- contraction not a multiple of the vector width, token count that is not Hp x Wp, feature width not a multiple of 4, odd debug
    knobs -- rather than launching."""
    from painter_amd._lib import PA_BF16, PA_F32, lib
    buf = torch.zeros(1 << 16, dtype=torch.uint8, device=DEV)
    p_, s = buf.data_ptr(), torch.cuda.current_stream().cuda_stream
    assert lib.pa_linear_fwd(PA_BF16, 0, p_, 12, p_, p_, p_, None, 16, None, None, 1, 4, 16, 12, s) != 0           # K % 8
    assert lib.pa_linear_fwd(PA_F32, 0, p_, 6, p_, p_, p_, None, 16, None, None, 1, 4, 16, 6, s) != 0              # K % 4
    assert lib.pa_attn_fwd(PA_BF16, p_, 192, p_, p_, 64, p_, None, 1, 100, 1, 8, 12, 64, 0.125, s) != 0             # L != Hp * Wp
    assert lib.pa_attn_fwd(PA_BF16, p_, 192, p_, p_, 64, p_, None, 1, 90, 1, 9, 10, 64, 0.125, s) != 0              # grid not a multiple of 4
    assert lib.pa_attn_fwd(PA_BF16, p_, 288, p_, p_, 96, p_, None, 1, 96, 1, 8, 12, 96, 0.125, s) != 0              # head_dim the kernels are not built for
    assert lib.pa_layernorm_fwd(PA_BF16, p_, 6, p_, p_, 1e-6, p_, 6, p_, p_, 4, 6, s) != 0                          # D % 4
    assert lib.pa_debug_set(99, 1) != 0
    assert lib.pa_attn_set_generation(7) != 0 and lib.pa_attn_set_generation(4) != 0        # the retired builds are gone
    with pytest.raises(RuntimeError):
        ops.linear_fwd(torch.zeros(4, 12, dtype=torch.bfloat16, device=DEV), torch.zeros(16, 12, dtype=torch.bfloat16, device=DEV), torch.zeros(16, device=DEV))
    torch.cuda.synchronize()
