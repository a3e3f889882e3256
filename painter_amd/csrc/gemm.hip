// nn.Linear forward / dgrad / wgrad entry points on the MFMA engine (gemm_engine.h).
// Replaces the reference's ATen addmm/mm calls at Painter/models_painter.py:76 (qkv), :87 (proj),
// timm Mlp fc1/fc2 (:201,:230), :423 (decoder_embed) and their autograd backward (SURVEY.md 8a a4,a9,a10,a13,a17).
#include <cstdlib>
#include "gemm_engine.h"
#include "gemm256.h"
#include "../../include/painter_hip.h"

// ------------------------------------------------------------------------------- epilogues
template <typename OutT> struct EpiBias {   // out[i][j] = acc + bias[j]
    OutT* out; size_t ldo; const float* bias; int M, N;
    DEVI void operator()(const f32x16 (&acc)[2][2], int ib, int jb, int lane, int) const {
        foreach_acc(acc, ib, jb, lane, [&](int i, int j, float v) {
            if (i < M && j < N) out[(size_t)i * ldo + j] = from_f<OutT>(v + (bias ? bias[j] : 0.f));
        });
    }
};
// pre = acc + bias (rounded to T); act = gelu(pre).  What the backward needs of `pre` is saved in `aux`: the pre-activation itself in the
// exact-fp32 build (the fc2 data gradient evaluates erf-GELU' on it to 1e-7); in the bf16 build gelu'(pre) -- the forward epilogue has
// Phi(pre) and exp(-pre^2 / 2) in registers anyway, and the fc2 data-gradient epilogue becomes one load and one multiply -- since round 6
// as an 8-BIT CODE q = round((g' + 0.13) * 255 / 1.26) (g' lies in [-0.129, 1.129]; decode g' = q * 1.26 / 255 - 0.13: absolute error
// <= 2.5e-3), uint8 [M, N] with a row pitch of ld BYTES: half the bytes of the bf16 form on the fc1 write AND on the fc2-dgrad read
// (2.4 GB per ViT-L step).  Measured before it shipped (profiles/r06_ab_gelu_aux_8bit.log): every parity gate
// unchanged, the fc1 / fc2 weight gradients' error against the reference's fp32 gradients unchanged to three digits, -0.29 ms per step.
constexpr float G8_OFF = 0.13f, G8_SCALE = 255.f / 1.26f, G8_STEP = 1.26f / 255.f;
DEVI uint32_t g8_code(float g) { return min((uint32_t)fmaf(g, G8_SCALE, G8_OFF * G8_SCALE + 0.5f), 255u); }      // v_cvt_u32_f32 truncates and saturates at 0
DEVI float g8_value(uint32_t q) { return fmaf((float)q, G8_STEP, -G8_OFF); }
template <typename T> struct GeluAux { typedef T type; };
template <> struct GeluAux<bf16> { typedef unsigned char type; };
template <typename T> struct EpiBiasGelu {
    typename GeluAux<T>::type* aux; T* act; size_t ld; const float* bias; int M, N;
    DEVI void operator()(const f32x16 (&acc)[2][2], int ib, int jb, int lane, int) const {
        foreach_acc(acc, ib, jb, lane, [&](int i, int j, float v) {
            if (i < M && j < N) {
                const T p = from_f<T>(v + bias[j]);
                if constexpr (std::is_same<T, bf16>::value) {
                    float c, e;
                    const float x = to_f(p);
                    gelu_parts(x, c, e);
                    if (aux) aux[(size_t)i * ld + j] = (unsigned char)g8_code(fmaf(x * 0.39894228040143268f, e, c));
                    act[(size_t)i * ld + j] = from_f<T>(x * c);
                } else {
                    if (aux) aux[(size_t)i * ld + j] = p;
                    act[(size_t)i * ld + j] = from_f<T>(gelu_f(to_f(p)));
                }
            }
        });
    }
};
struct EpiBiasResid {   // out = resid + rowscale[i / rps] * (acc + bias)   (residual add + DropPath factor)
    float* out; const float* resid; size_t ld; const float* bias; const float* rowscale; int rps; int M, N;
    DEVI void operator()(const f32x16 (&acc)[2][2], int ib, int jb, int lane, int) const {
        foreach_acc(acc, ib, jb, lane, [&](int i, int j, float v) {
            if (i < M && j < N) {
                const float s = rowscale ? rowscale[i / rps] : 1.f;
                out[(size_t)i * ld + j] = resid[(size_t)i * ld + j] + s * (v + bias[j]);
            }
        });
    }
};
template <typename T> struct EpiDGelu {   // out = acc * gelu'(pre); aux = what EpiBiasGelu<T> saved (fp32: pre, bf16: the 8-bit code of gelu'(pre))
    T* out; const typename GeluAux<T>::type* aux; size_t ld; int M, N;
    DEVI void operator()(const f32x16 (&acc)[2][2], int ib, int jb, int lane, int) const {
        foreach_acc(acc, ib, jb, lane, [&](int i, int j, float v) {
            if (i < M && j < N) {
                if constexpr (std::is_same<T, bf16>::value) out[(size_t)i * ld + j] = from_f<T>(v * g8_value(aux[(size_t)i * ld + j]));
                else out[(size_t)i * ld + j] = from_f<T>(v * gelu_grad_f(aux[(size_t)i * ld + j]));
            }
        });
    }
};
template <typename T> struct EpiPixShuf {   // token-major [b,h,w][p,q,c] -> NHWC image (models_painter.py:424-428)
    T* out; const float* bias; int Hp, Wp, P, C, M, N;
    DEVI void operator()(const f32x16 (&acc)[2][2], int ib, int jb, int lane, int) const {
        const int L = Hp * Wp;
        foreach_acc(acc, ib, jb, lane, [&](int i, int j, float v) {
            if (i < M && j < N) {
                const int b = i / L, l = i % L, h = l / Wp, w = l % Wp;
                const int c = j % C, pq = j / C, q = pq % P, p = pq / P;
                const size_t y = (size_t)b * Hp * P + h * P + p, x = (size_t)w * P + q;
                out[(y * (size_t)(Wp * P) + x) * C + c] = from_f<T>(v + bias[j]);
            }
        });
    }
};
struct EpiSlab {   // fp32 partial result of split z (wgrad)
    float* out; size_t ldo; size_t slab; int M, N;
    DEVI void operator()(const f32x16 (&acc)[2][2], int ib, int jb, int lane, int z) const {
        float* o = out + (size_t)z * slab;
        foreach_acc(acc, ib, jb, lane, [&](int i, int j, float v) {
            if (i < M && j < N) o[(size_t)i * ldo + j] = v;
        });
    }
};

// ---- 8-wide epilogues of the 256x256 bf16 kernel (gemm256.h): (row i, column j % 8 == 0, D[i][j..j+3], D[i][j+4..j+7], split)
DEVI void store8(bf16* p, float4 a, float4 b) {
    *reinterpret_cast<uint4*>(p) = make_uint4(pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w), pack_bf16x2(b.x, b.y), pack_bf16x2(b.z, b.w));
}
DEVI void store8(float* p, float4 a, float4 b) {
    *reinterpret_cast<float4*>(p) = a;
    *reinterpret_cast<float4*>(p + 4) = b;
}
// fp32 outputs in the split column layout (gemm256.h, epi_hi_off): columns j..j+3 and j+32..j+35, each half bounds-checked on its own
// (N % 8 == 0 and j % 4 == 0: a half that starts inside the matrix ends inside it)
DEVI void store44(float* p, int j, int N, float4 a, float4 b) {
    if (j < N) *reinterpret_cast<float4*>(p) = a;
    if (j + 32 < N) *reinterpret_cast<float4*>(p + 32) = b;
}
DEVI float4 load4(const float* p) { return *reinterpret_cast<const float4*>(p); }
DEVI float4 add4(float4 a, float4 b) {       // two v_pk_add_f32 (packed fp32 is exact: common.h, gelu_parts2)
    const f32x2_t l = (f32x2_t){a.x, a.y} + (f32x2_t){b.x, b.y}, h = (f32x2_t){a.z, a.w} + (f32x2_t){b.z, b.w};
    return make_float4(l[0], l[1], h[0], h[1]);
}
DEVI void load8(const bf16* p, float4& a, float4& b) {
    const uint4 w = *reinterpret_cast<const uint4*>(p);
    a = make_float4(bf16_lo(w.x), bf16_hi(w.x), bf16_lo(w.y), bf16_hi(w.y));
    b = make_float4(bf16_lo(w.z), bf16_hi(w.z), bf16_lo(w.w), bf16_hi(w.w));
}
struct EpiNone {};
struct EpiCol8 { float4 a, b; };          // bias[j..j+3], bias[j+4..j+7]
DEVI EpiCol8 load_col8(const float* bias, int j, int N) {
    EpiCol8 c{make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0)};
    if (bias != nullptr && j < N) { c.a = load4(bias + j); c.b = load4(bias + j + 4); }
    return c;
}
DEVI EpiCol8 load_col44(const float* bias, int j, int N) {      // split layout: bias[j..j+3], bias[j+32..j+35]
    EpiCol8 c{make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0)};
    if (bias != nullptr && j < N) c.a = load4(bias + j);
    if (bias != nullptr && j + 32 < N) c.b = load4(bias + j + 32);
    return c;
}
template <typename OutT> struct Epi4Bias {
    OutT* out; size_t ldo; const float* bias; int M, N;
    static constexpr int HI_OFF = std::is_same<OutT, float>::value ? 32 : 4;
    typedef EpiCol8 Col;
    typedef EpiNone Row;
    DEVI Col col(int j) const { return HI_OFF == 32 ? load_col44(bias, j, N) : load_col8(bias, j, N); }
    DEVI Row row(int, int) const { return Row{}; }
    DEVI void store(int i, int j, float4 a, float4 b, const Col& c, const Row&, int) const {
        if (i >= M || j >= N) return;
        if constexpr (std::is_same<OutT, float>::value) store44(out + (size_t)i * ldo + j, j, N, add4(a, c.a), add4(b, c.b));
        else store8(out + (size_t)i * ldo + j, add4(a, c.a), add4(b, c.b));
    }
};
struct Epi4BiasGelu {       // aux (optional): the 8-bit code of gelu'(pre) (see EpiBiasGelu), row pitch ld bytes
    unsigned char* aux; bf16* act; size_t ld; const float* bias; int M, N;
    typedef EpiCol8 Col;
    typedef EpiNone Row;
    DEVI Col col(int j) const { return load_col8(bias, j, N); }
    DEVI Row row(int, int) const { return Row{}; }
    DEVI void store(int i, int j, float4 a, float4 b, const Col& c, const Row&, int) const {
        if (i >= M || j >= N) return;
        a = add4(a, c.a);
        b = add4(b, c.b);
        // GELU and GELU' are taken of the bf16-ROUNDED pre-activation, as the reference's autocast path does (its fc1 output is a bf16 tensor)
        const uint4 pk = make_uint4(pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w), pack_bf16x2(b.x, b.y), pack_bf16x2(b.z, b.w));
        if (aux) {
            f32x2_t g0, g1, g2, g3, d0, d1, d2, d3;
            gelu_both2(bf16_lo(pk.x), bf16_hi(pk.x), g0, d0);
            gelu_both2(bf16_lo(pk.y), bf16_hi(pk.y), g1, d1);
            gelu_both2(bf16_lo(pk.z), bf16_hi(pk.z), g2, d2);
            gelu_both2(bf16_lo(pk.w), bf16_hi(pk.w), g3, d3);
            const uint32_t lo = g8_code(d0[0]) | (g8_code(d0[1]) << 8) | (g8_code(d1[0]) << 16) | (g8_code(d1[1]) << 24);
            const uint32_t hi = g8_code(d2[0]) | (g8_code(d2[1]) << 8) | (g8_code(d3[0]) << 16) | (g8_code(d3[1]) << 24);
            *reinterpret_cast<uint2*>(aux + (size_t)i * ld + j) = make_uint2(lo, hi);
            store8(act + (size_t)i * ld + j, make_float4(g0[0], g0[1], g1[0], g1[1]), make_float4(g2[0], g2[1], g3[0], g3[1]));
        } else {
            const f32x2_t g0 = gelu_fast2(bf16_lo(pk.x), bf16_hi(pk.x)), g1 = gelu_fast2(bf16_lo(pk.y), bf16_hi(pk.y)),
                          g2 = gelu_fast2(bf16_lo(pk.z), bf16_hi(pk.z)), g3 = gelu_fast2(bf16_lo(pk.w), bf16_hi(pk.w));
            store8(act + (size_t)i * ld + j, make_float4(g0[0], g0[1], g1[0], g1[1]), make_float4(g2[0], g2[1], g3[0], g3[1]));
        }
    }
};
struct Epi4BiasResid {
    float* out; const float* resid; size_t ld; const float* bias; const float* rowscale; int rps; int M, N;
    static constexpr int HI_OFF = 32;        // fp32 read-modify-write: every residual load and every store covers whole 128-byte segments
    typedef EpiCol8 Col;
    struct Row { float4 ra, rb; float s; };
    DEVI Col col(int j) const { return load_col44(bias, j, N); }
    DEVI Row row(int i, int j) const {
        Row r{make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0), 1.f};
        if (i < M && j < N) {
            r.ra = load4(resid + (size_t)i * ld + j);
            if (j + 32 < N) r.rb = load4(resid + (size_t)i * ld + j + 32);
            if (rowscale) r.s = rowscale[i / rps];
        }
        return r;
    }
    DEVI void store(int i, int j, float4 a, float4 b, const Col& c, const Row& r, int) const {
        if (i >= M || j >= N) return;
        const float s = r.s;
        a = add4(a, c.a);
        b = add4(b, c.b);
        store44(out + (size_t)i * ld + j, j, N, make_float4(r.ra.x + s * a.x, r.ra.y + s * a.y, r.ra.z + s * a.z, r.ra.w + s * a.w),
                make_float4(r.rb.x + s * b.x, r.rb.y + s * b.y, r.rb.z + s * b.z, r.rb.w + s * b.w));
    }
};
struct Epi4DGelu {          // out = acc * gelu'(pre), gelu'(pre) from the 8-bit code the fc1 forward epilogue saved (Epi4BiasGelu); row pitches: out ld elements, aux ld bytes
    bf16* out; const unsigned char* aux; size_t ld; int M, N;
    typedef EpiNone Col;
    struct Row { uint2 p; };
    DEVI Col col(int) const { return Col{}; }
    DEVI Row row(int i, int j) const {
        Row r{make_uint2(0, 0)};
        if (i < M && j < N) r.p = *reinterpret_cast<const uint2*>(aux + (size_t)i * ld + j);
        return r;
    }
    DEVI void decode(const Row& r, float (&g)[8]) const {
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] = g8_value(((e < 4 ? r.p.x : r.p.y) >> (8 * (e & 3))) & 0xffu);
    }
    DEVI void store(int i, int j, float4 a, float4 b, const Col&, const Row& r, int) const {
        if (i >= M || j >= N) return;
        float g[8];
        decode(r, g);
        store8(out + (size_t)i * ld + j, make_float4(a.x * g[0], a.y * g[1], a.z * g[2], a.w * g[3]), make_float4(b.x * g[4], b.y * g[5], b.z * g[6], b.w * g[7]));
    }
};
// the same with the column sums of dX (the fp32 values in front of its bf16 rounding; gemm256.h, epi_colsum): part f32 [2 * row tiles][N]
struct Epi4DGeluCS : Epi4DGelu {
    float* part;
    DEVI void store_cs(int i, int j, float4 a, float4 b, const Col&, const Row& r, int, float (&cs)[8]) const {
        if (i >= M || j >= N) return;
        float g[8];
        decode(r, g);
        const uint4 pk = make_uint4(pack_bf16x2(a.x * g[0], a.y * g[1]), pack_bf16x2(a.z * g[2], a.w * g[3]),
                                    pack_bf16x2(b.x * g[4], b.y * g[5]), pack_bf16x2(b.z * g[6], b.w * g[7]));
        *reinterpret_cast<uint4*>(out + (size_t)i * ld + j) = pk;
        cs[0] = fmaf(a.x, g[0], cs[0]); cs[1] = fmaf(a.y, g[1], cs[1]); cs[2] = fmaf(a.z, g[2], cs[2]); cs[3] = fmaf(a.w, g[3], cs[3]);      // the fp32 values in front of the rounding
        cs[4] = fmaf(b.x, g[4], cs[4]); cs[5] = fmaf(b.y, g[5], cs[5]); cs[6] = fmaf(b.z, g[6], cs[6]); cs[7] = fmaf(b.w, g[7], cs[7]);
    }
    DEVI void colsum_out(int prow, int j, const float (&cs)[8]) const {
        if (j >= N) return;
        float* o = part + (size_t)prow * N + j;
        *reinterpret_cast<float4*>(o) = make_float4(cs[0], cs[1], cs[2], cs[3]);
        *reinterpret_cast<float4*>(o + 4) = make_float4(cs[4], cs[5], cs[6], cs[7]);
    }
};
template <> struct g256::epi_colsum<Epi4DGeluCS> { static constexpr bool value = true; };
struct Epi4PixShuf {
    bf16* out; const float* bias; int Hp, Wp, P, C, M, N;
    typedef EpiCol8 Col;
    typedef EpiNone Row;
    DEVI Col col(int j) const { return load_col8(bias, j, N); }
    DEVI Row row(int, int) const { return Row{}; }
    DEVI void store(int i, int j, float4 a, float4 b, const Col& cc, const Row&, int) const {
        if (i >= M || j >= N) return;
        const int L = Hp * Wp;
        const int bb = i / L, l = i - bb * L, h = l / Wp, w = l - h * Wp;
        const int c = j % C, pq = j / C, q = pq % P, p = pq / P;
        const size_t y = (size_t)bb * Hp * P + h * P + p, x = (size_t)w * P + q;
        store8(out + (y * (size_t)(Wp * P) + x) * C + c, add4(a, cc.a), add4(b, cc.b));
    }
};
struct Epi4Slab {
    float* out; size_t ldo; size_t slab; int M, N;
    static constexpr int HI_OFF = 32;
    typedef EpiNone Col;
    typedef EpiNone Row;
    DEVI Col col(int) const { return Col{}; }
    DEVI Row row(int, int) const { return Row{}; }
    DEVI void store(int i, int j, float4 a, float4 b, const Col&, const Row&, int split) const {
        if (i >= M || j >= N) return;
        store44(out + (size_t)split * slab + (size_t)i * ldo + j, j, N, a, b);
    }
};

// ------------------------------------------------------------------------------- reductions
// out[n] = (accumulate ? out[n] : 0) + sum_z in[z * stride + n]      (n % 4 == 0)
// block = 16 column-groups (float4) x 16 z-lanes; the z loop is strided over the z-lanes and unrolled so that many loads
// are in flight, then the 16 partial sums are combined through LDS in a fixed order (deterministic).
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ in, float* out, size_t n, int nz, size_t stride, int accumulate) {
    __shared__ float4 sh[16][17];
    const int cg = threadIdx.x & 15, zl = threadIdx.x >> 4;
    const size_t i = ((size_t)blockIdx.x * 16 + cg) * 4;
    float4 s = make_float4(0, 0, 0, 0);
    if (i < n) {
#pragma unroll 4
        for (int z = zl; z < nz; z += 16) {
            const float4 v = *reinterpret_cast<const float4*>(in + (size_t)z * stride + i);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    sh[zl][cg] = s;
    __syncthreads();
    if (zl == 0 && i < n) {
        float4 t = accumulate ? *reinterpret_cast<const float4*>(out + i) : make_float4(0, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < 16; ++k) { const float4 v = sh[k][cg]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        *reinterpret_cast<float4*>(out + i) = t;
    }
}
// few slabs, large n: one float4 per thread, plain loop
__global__ void slab_reduce_wide_kernel(const float* __restrict__ in, float* out, size_t n, int nz, size_t stride, int accumulate) {
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    float4 s = accumulate ? *reinterpret_cast<const float4*>(out + i) : make_float4(0, 0, 0, 0);
#pragma unroll 4
    for (int z = 0; z < nz; ++z) {
        const float4 v = *reinterpret_cast<const float4*>(in + (size_t)z * stride + i);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    *reinterpret_cast<float4*>(out + i) = s;
}
// the same reduction with the column range split over two destinations: columns [0, n0) -> out0, [n0, n) -> out1 (LayerNorm backward:
// [dgamma | dbeta] and the bias column sum from one set of partial rows)
__global__ __launch_bounds__(256) void slab_reduce2_kernel(const float* __restrict__ in, float* out0, float* out1, size_t n0, size_t n, int nz, size_t stride) {
    __shared__ float4 sh[16][17];
    const int cg = threadIdx.x & 15, zl = threadIdx.x >> 4;
    const size_t i = ((size_t)blockIdx.x * 16 + cg) * 4;
    float4 s = make_float4(0, 0, 0, 0);
    if (i < n) {
#pragma unroll 4
        for (int z = zl; z < nz; z += 16) {
            const float4 v = *reinterpret_cast<const float4*>(in + (size_t)z * stride + i);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    sh[zl][cg] = s;
    __syncthreads();
    if (zl == 0 && i < n) {
        float4 t = make_float4(0, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < 16; ++k) { const float4 v = sh[k][cg]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        *reinterpret_cast<float4*>(i < n0 ? out0 + i : out1 + (i - n0)) = t;
    }
}
int pa_slab_reduce2(const float* in, float* out0, float* out1, int64_t n0, int64_t n, int nz, int64_t stride, hipStream_t st) {
    if (n <= 0) return 0;
    if (n % 4 || n0 % 4 || stride % 4 || (n > n0 && out1 == nullptr)) return (int)hipErrorInvalidValue;
    PA_LAUNCH(slab_reduce2_kernel, dim3((unsigned)((n / 4 + 15) / 16)), dim3(256), 0, st, in, out0, out1, (size_t)n0, (size_t)n, nz, (size_t)stride);
    LAUNCH_CHECK();
}
extern "C" int pa_slab_reduce(const float* in, float* out, int64_t n, int nz, int64_t stride, int accumulate, hipStream_t st) {
    if (n <= 0) return 0;
    if (n % 4 || stride % 4) return (int)hipErrorInvalidValue;
    if (nz <= 16 && n >= (1 << 18))
        PA_LAUNCH(slab_reduce_wide_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, in, out, (size_t)n, nz, (size_t)stride, accumulate);
    else
        PA_LAUNCH(slab_reduce_kernel, dim3((unsigned)((n / 4 + 15) / 16)), dim3(256), 0, st, in, out, (size_t)n, nz, (size_t)stride, accumulate);
    LAUNCH_CHECK();
}

// column sums of a [M, N] T matrix (bias gradients): stage 1 partial[chunk][N], stage 2 slab reduce (fixed order)
// block = 32 column-lanes (8 consecutive columns each, one 16-byte load for bf16) x 8 row-lanes; 128 rows per block
template <typename T> __global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, size_t ld, int M, int N, int rows_per_chunk, float* __restrict__ part) {
    const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int c0 = (blockIdx.x * 32 + cl) * 8;
    const int r0 = blockIdx.y * rows_per_chunk, r1 = min(M, r0 + rows_per_chunk);
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (c0 < N) {
#pragma unroll 4
        for (int r = r0 + rl; r < r1; r += 8) {
            const T* p = x + (size_t)r * ld + c0;
            if constexpr (sizeof(T) == 2) {
                const uint4 w = *reinterpret_cast<const uint4*>(p);
                s[0] += bf16_lo(w.x); s[1] += bf16_hi(w.x); s[2] += bf16_lo(w.y); s[3] += bf16_hi(w.y);
                s[4] += bf16_lo(w.z); s[5] += bf16_hi(w.z); s[6] += bf16_lo(w.w); s[7] += bf16_hi(w.w);
            } else {
                const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
                s[0] += a.x; s[1] += a.y; s[2] += a.z; s[3] += a.w; s[4] += b.x; s[5] += b.y; s[6] += b.z; s[7] += b.w;
            }
        }
    }
    __shared__ float red[8][32 * 8 + 8];
#pragma unroll
    for (int e = 0; e < 8; ++e) red[rl][cl * 8 + e] = s[e];
    __syncthreads();
    const int c = threadIdx.x;             // 256 columns of this block
    if (blockIdx.x * 256 + c < N) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red[k][c];
        part[(size_t)blockIdx.y * N + blockIdx.x * 256 + c] = t;
    }
}
static int colsum_rows_per_chunk(int M) { return M >= 4096 ? 128 : 32; }
extern "C" int64_t pa_colsum_workspace_bytes(int M, int N) {
    const int rpc = colsum_rows_per_chunk(M), chunks = (M + rpc - 1) / rpc;
    return (int64_t)chunks * N * sizeof(float);
}
extern "C" int pa_colsum(int dtype, const void* x, int64_t ld, int M, int N, float* out, void* workspace, hipStream_t st) {
    if (N % 8 || ld % 8) return (int)hipErrorInvalidValue;
    const int rpc = colsum_rows_per_chunk(M), chunks = (M + rpc - 1) / rpc;
    float* part = reinterpret_cast<float*>(workspace);
    dim3 grid((N + 255) / 256, chunks);
    if (dtype == PA_BF16)
        PA_LAUNCH(colsum_kernel<bf16>, grid, dim3(256), 0, st, (const bf16*)x, (size_t)ld, M, N, rpc, part);
    else
        PA_LAUNCH(colsum_kernel<float>, grid, dim3(256), 0, st, (const float*)x, (size_t)ld, M, N, rpc, part);
    int e = (int)hipGetLastError();
    if (e) return e;
    return pa_slab_reduce(part, out, N, chunks, N, 0, st);
}

// ------------------------------------------------------------------------------- linear forward
template <typename T>
static int linear_fwd_t(int epi, const T* x, int64_t ldx, const T* w, const float* bias, void* out, void* out2,
                        int64_t ldo, const float* resid, const float* rowscale, int rps, int M, int N, int K,
                        hipStream_t st) {
    if constexpr (std::is_same<T, bf16>::value) {
        if (g256::ok(M, N, K, false, false, ldx, K)) {
            switch (epi) {
            case PA_EPI_BIAS:
                return g256::launch<false, false>(x, ldx, w, K, Epi4Bias<bf16>{(bf16*)out, (size_t)ldo, bias, M, N}, M, N, K, 1, st);
            case PA_EPI_BIAS_F32:
                return g256::launch<false, false>(x, ldx, w, K, Epi4Bias<float>{(float*)out, (size_t)ldo, bias, M, N}, M, N, K, 1, st);
            case PA_EPI_BIAS_GELU:
                return g256::launch<false, false>(x, ldx, w, K, Epi4BiasGelu{(unsigned char*)out2, (bf16*)out, (size_t)ldo, bias, M, N}, M, N, K, 1, st);
            case PA_EPI_BIAS_RESID:
                return g256::launch<false, false>(x, ldx, w, K, Epi4BiasResid{(float*)out, resid, (size_t)ldo, bias, rowscale, rps, M, N}, M, N, K, 1, st);
            }
        }
    }
    OpN<T> A{x, (size_t)ldx, M, 0};
    OpN<T> B{w, (size_t)K, N, 0};
    switch (epi) {
    case PA_EPI_BIAS:
        return launch_gemm<T, 2, 2>(A, B, EpiBias<T>{(T*)out, (size_t)ldo, bias, M, N}, M, N, K, 1, 1, st);
    case PA_EPI_BIAS_F32:
        return launch_gemm<T, 2, 2>(A, B, EpiBias<float>{(float*)out, (size_t)ldo, bias, M, N}, M, N, K, 1, 1, st);
    case PA_EPI_BIAS_GELU:
        return launch_gemm<T, 2, 2>(A, B, EpiBiasGelu<T>{(typename GeluAux<T>::type*)out2, (T*)out, (size_t)ldo, bias, M, N}, M, N, K, 1, 1, st);
    case PA_EPI_BIAS_RESID:
        return launch_gemm<T, 2, 2>(A, B, EpiBiasResid{(float*)out, resid, (size_t)ldo, bias, rowscale, rps, M, N}, M, N, K, 1, 1, st);
    }
    return (int)hipErrorInvalidValue;
}

extern "C" int pa_linear_fwd(int dtype, int epilogue, const void* x, int64_t ldx, const void* w, const float* bias,
                             void* out, void* out2, int64_t ldo, const float* resid, const float* rowscale,
                             int rows_per_sample, int M, int N, int K, hipStream_t st) {
    if (K % (dtype == PA_BF16 ? 8 : 4)) return (int)hipErrorInvalidValue;
    if (dtype == PA_BF16)
        return linear_fwd_t<bf16>(epilogue, (const bf16*)x, ldx, (const bf16*)w, bias, out, out2, ldo, resid, rowscale, rows_per_sample, M, N, K, st);
    return linear_fwd_t<float>(epilogue, (const float*)x, ldx, (const float*)w, bias, out, out2, ldo, resid, rowscale, rows_per_sample, M, N, K, st);
}

template <typename T>
static int linear_pixshuf_t(const T* x, int64_t ldx, const T* w, const float* bias, T* out, int Bn, int Hp, int Wp,
                            int P, int C, int K, hipStream_t st) {
    const int M = Bn * Hp * Wp, N = P * P * C;
    if constexpr (std::is_same<T, bf16>::value) {
        if (g256::ok(M, N, K, false, false, ldx, K) && C % 8 == 0)
            return g256::launch<false, false>(x, ldx, w, K, Epi4PixShuf{out, bias, Hp, Wp, P, C, M, N}, M, N, K, 1, st);
    }
    OpN<T> A{x, (size_t)ldx, M, 0};
    OpN<T> B{w, (size_t)K, N, 0};
    return launch_gemm<T, 2, 2>(A, B, EpiPixShuf<T>{out, bias, Hp, Wp, P, C, M, N}, M, N, K, 1, 1, st);
}
extern "C" int pa_linear_pixshuf(int dtype, const void* x, int64_t ldx, const void* w, const float* bias, void* out_nhwc,
                                 int batch, int Hp, int Wp, int P, int C, int K, hipStream_t st) {
    if (K % 8) return (int)hipErrorInvalidValue;
    if (dtype == PA_BF16) return linear_pixshuf_t<bf16>((const bf16*)x, ldx, (const bf16*)w, bias, (bf16*)out_nhwc, batch, Hp, Wp, P, C, K, st);
    return linear_pixshuf_t<float>((const float*)x, ldx, (const float*)w, bias, (float*)out_nhwc, batch, Hp, Wp, P, C, K, st);
}

// ------------------------------------------------------------------------------- linear backward
// dX[M,K] = dY[M,N] . W[N,K]      (contraction over N; W is contraction-major -> OpT)
// dx_colsum (optional): f32 [K] = column sums of dX as stored -- partial rows from the GEMM's epilogue on the bf16 fast path when the
// GELU side input is given (workspace), a separate pa_colsum pass over dX otherwise
extern "C" int64_t pa_linear_dgrad_workspace_bytes(int M, int K) {
    const int64_t fast = (int64_t)2 * ((M + g256::BM_HALF - 1) / g256::BM_HALF) * K * sizeof(float);      // (an upper bound for every tile plan: all rows as half tiles)
    const int64_t slow = pa_colsum_workspace_bytes(M, K);
    return fast > slow ? fast : slow;
}
template <typename T>
static int linear_dgrad_t(const T* dy, int64_t lddy, const T* w, const typename GeluAux<T>::type* pre /* = gelu_aux: EpiBiasGelu */, T* dx, int64_t lddx, float* dx_colsum, float* ws, int M,
                          int N, int K, hipStream_t st) {
    if constexpr (std::is_same<T, bf16>::value) {
        if (g256::ok(M, K, N, false, true, lddy, K)) {
            if (pre && dx_colsum) {
                Epi4DGeluCS ep;
                ep.out = dx; ep.aux = pre; ep.ld = (size_t)lddx; ep.M = M; ep.N = K; ep.part = ws;
                int e = g256::launch<false, true>(dy, lddy, w, K, ep, M, K, N, 1, st);
                if (e) return e;
                return pa_slab_reduce(ws, dx_colsum, K, 2 * g256::row_tiles_used(M, K, 1), K, 0, st);
            }
            if (pre) return g256::launch<false, true>(dy, lddy, w, K, Epi4DGelu{dx, pre, (size_t)lddx, M, K}, M, K, N, 1, st);
            int e = g256::launch<false, true>(dy, lddy, w, K, Epi4Bias<bf16>{dx, (size_t)lddx, nullptr, M, K}, M, K, N, 1, st);
            if (e || dx_colsum == nullptr) return e;
            return pa_colsum(PA_BF16, dx, lddx, M, K, dx_colsum, ws, st);      // no GELU side input: a pass over the stored dX (the epilogue sums only exist beside it)
        }
    }
    OpN<T> A{dy, (size_t)lddy, M, 0};
    OpT<T> B{w, (size_t)K, K, 0};
    int e;
    if (pre) e = launch_gemm<T, 2, 2>(A, B, EpiDGelu<T>{dx, pre, (size_t)lddx, M, K}, M, K, N, 1, 1, st);
    else e = launch_gemm<T, 2, 2>(A, B, EpiBias<T>{dx, (size_t)lddx, nullptr, M, K}, M, K, N, 1, 1, st);
    if (e || dx_colsum == nullptr) return e;
    return pa_colsum(std::is_same<T, bf16>::value ? PA_BF16 : PA_F32, dx, lddx, M, K, dx_colsum, ws, st);
}
extern "C" int pa_linear_dgrad(int dtype, const void* dy, int64_t lddy, const void* w, const void* gelu_aux,
                               void* dx, int64_t lddx, float* dx_colsum, void* workspace, int M, int N, int K, hipStream_t st) {
    if (N % 8 || K % 4) return (int)hipErrorInvalidValue;
    if (dx_colsum != nullptr && (workspace == nullptr || K % 8 || lddx % 8)) return (int)hipErrorInvalidValue;
    if (dtype == PA_BF16) return linear_dgrad_t<bf16>((const bf16*)dy, lddy, (const bf16*)w, (const unsigned char*)gelu_aux, (bf16*)dx, lddx, dx_colsum, (float*)workspace, M, N, K, st);
    return linear_dgrad_t<float>((const float*)dy, lddy, (const float*)w, (const float*)gelu_aux, (float*)dx, lddx, dx_colsum, (float*)workspace, M, N, K, st);
}

// dW[N,K] = dY[M,N]^T . X[M,K]    (contraction over M; both operands contraction-major), split-K + reduce
static bool wgrad_fast(int dtype, int M, int N, int K) {
    return dtype == PA_BF16 && g256::ok(N, K, M, true, true, N, K) && K % 4 == 0 && N % 256 == 0 && K % 256 == 0;
}
static int wgrad_fast_splits(int M, int N, int K) {
    const int tiles = (N / 256) * (K / 256);
    // target number of workgroups: 256 fills the chip when the kernel runs alone.  On the side stream, beside the data-gradient
    // chain, fewer and longer workgroups are better (half the slab traffic, and the kernel does not need the whole chip):
    // interleaved A/B on one box: 256 -> 132.9 / 133.1 images/s, 128 -> 135.0 / 134.9, 64 -> 136.1 (with 64 the fc1/fc2/qkv weight
    // gradients need no K split at all).  pa_debug_set(3, n) / PA_WGRAD_WGS select it.
    static const int env_target = [] { const char* v = getenv("PA_WGRAD_WGS"); return v ? atoi(v) : 0; }();
    int target = env_target > 0 ? env_target : (g256::g_dbg[3] > 0 ? g256::g_dbg[3] : 256);
    if (target < 16) target = 16;
    int s = (target + 3 * tiles / 4) / tiles;          // (rounds up from x.25: ViT-H/14's 75-tile qkv gradient gets 2 splits = 150 workgroups at target 96, not 75 long ones.
    if (s < 1) s = 1;                                   //  At the side-stream target 96 every ViT-L shape -- 16 / 48 / 64 tiles -- keeps the count round-half-up gave it; at the
                                                        //  stand-alone target 256 (one stream, bench.py's profiled pass) the 48-tile qkv gradient goes from 5 to 6 splits, so the
                                                        //  profiled gemm256_wgrad / slab figures from round 5 on are not like-for-like with profiles/r0[1-4]_*.csv)
    const int ktiles = M / 64;
    if (s > ktiles / 4) s = ktiles / 4 > 0 ? ktiles / 4 : 1;
    return g256::splits_used(M, s);
}
static int wgrad_splits(int M, int N, int K, int bk) {
    const int tiles = ((N + 127) / 128) * ((K + 127) / 128);
    const int nku = (M + bk - 1) / bk;
    int s = (640 + tiles - 1) / tiles;        // aim for >= 2.5 workgroups per CU
    if (s > nku) s = nku;
    if (s > 64) s = 64;
    if (s < 1) s = 1;
    return s;
}
extern "C" int64_t pa_linear_wgrad_workspace_bytes(int dtype, int M, int N, int K) {
    if (wgrad_fast(dtype, M, N, K)) {
        const int s = wgrad_fast_splits(M, N, K);
        return s > 1 ? (int64_t)s * N * K * sizeof(float) : 0;
    }
    const int s = wgrad_splits(M, N, K, dtype == PA_BF16 ? 64 : 32);
    return s > 1 ? (int64_t)s * N * K * sizeof(float) : 0;
}
template <typename T>
static int linear_wgrad_t(const T* dy, int64_t lddy, const T* x, int64_t ldx, float* dw, float* ws, int M, int N, int K,
                          hipStream_t st) {
    if constexpr (std::is_same<T, bf16>::value) {
        if (wgrad_fast(PA_BF16, M, N, K) && g256::ok(N, K, M, true, true, lddy, ldx)) {
            const int s = wgrad_fast_splits(M, N, K);
            if (s == 1) return g256::launch<true, true>(dy, lddy, x, ldx, Epi4Slab{dw, (size_t)K, 0, N, K}, N, K, M, 1, st);
            int e = g256::launch<true, true>(dy, lddy, x, ldx, Epi4Slab{ws, (size_t)K, (size_t)N * K, N, K}, N, K, M, s, st);
            if (e) return e;
            return pa_slab_reduce(ws, dw, (int64_t)N * K, s, (int64_t)N * K, 0, st);
        }
    }
    OpT<T> A{dy, (size_t)lddy, N, 0};
    OpT<T> B{x, (size_t)ldx, K, 0};
    const int s = wgrad_splits(M, N, K, TT<T>::BK);
    if (s == 1) return launch_gemm<T, 2, 2>(A, B, EpiSlab{dw, (size_t)K, 0, N, K}, N, K, M, 1, 1, st);
    int e = launch_gemm<T, 2, 2>(A, B, EpiSlab{ws, (size_t)K, (size_t)N * K, N, K}, N, K, M, s, 1, st);
    if (e) return e;
    return pa_slab_reduce(ws, dw, (int64_t)N * K, s, (int64_t)N * K, 0, st);
}
extern "C" int pa_linear_wgrad(int dtype, const void* dy, int64_t lddy, const void* x, int64_t ldx, float* dw,
                               void* workspace, int M, int N, int K, hipStream_t st) {
    if (N % 4 || K % 4) return (int)hipErrorInvalidValue;
    if (dtype == PA_BF16) return linear_wgrad_t<bf16>((const bf16*)dy, lddy, (const bf16*)x, ldx, dw, (float*)workspace, M, N, K, st);
    return linear_wgrad_t<float>((const float*)dy, lddy, (const float*)x, ldx, dw, (float*)workspace, M, N, K, st);
}

extern "C" int pa_abi_version(void) { return PA_ABI_VERSION; }
// knobs: 0-4 gemm256 (g256::g_dbg), 5 the G256_ILV_AB schedule override (experiment builds), 6 rel-pos splits, 7 fused rel-pos gradient,
// 8 light attention workgroups last, 9 conv3x3 weight-gradient groups, 10 LayerNorm-backward variant (round 5 shared index 5 with the ILV
// override: tools that swept one silently switched the other)
extern "C" int pa_debug_get(int which) {
    if (which < 0 || which > 15) return -1;
    if (which == 9) return g_conv_wgrad_groups;
    if (which == 10) return g_ln_bwd_variant;
    if (which > 10) return g_misc_knob[which - 11];
    if (which == 6) return g_relpos_splits;
    if (which == 7) return g_attn3_fuse;
    if (which == 8) return g_attn_light_last;
    return g256::g_dbg[which];
}
extern "C" int pa_debug_set(int which, int value) {
    if (which < 0 || which > 15) return (int)hipErrorInvalidValue;
    if (which < 8) g256::g_dbg[which] = value;
    if (which == 9) g_conv_wgrad_groups = value;
    if (which == 10) g_ln_bwd_variant = value;
    if (which > 10) g_misc_knob[which - 11] = value;
    if (which == 6) g_relpos_splits = value;
    if (which == 7) g_attn3_fuse = value;
    if (which == 8) g_attn_light_last = value;
    return 0;
}
