// Gather-operand contractions on the MFMA engine:
//   * PatchEmbed conv k16 s16 (+ token assembly) -- Painter/util/vitdet_utils.py:182-186, models_painter.py:387-409,
//     SegGPT type tokens models_seggpt.py:415-420  
//   * decoder_pred tail: Conv3x3(64->64) -> LayerNorm2D(64) -> GELU -> Conv1x1(64->3), fused in one kernel
//     (models_painter.py:328-333, :430; util/vitdet_utils.py:204-209)  
//   * their weight / data gradients.
#include "gemm_engine.h"
#include "gemm256.h"
#include "conv64.h"
#include "../../include/painter_hip.h"

// ------------------------------------------------------------------------------- patch embed operands
// A(row = token t in [0, S*B*L), k = c*P*P + ph*P + pw) = img_s[b, c, h*P+ph, w*P+pw]   (fp32 NCHW source)
// GEN = false: P % 8 == 0, a 16-byte operand chunk is 8 (bf16) / 4 (fp32) consecutive pixels of one patch row, loaded as float4s.
// GEN = true : any P (ViT-H/14: P = 14, K = 588): element-wise gather, k >= kreal (the zero padding of K up to a multiple of 8) reads 0.
template <typename T, bool GEN = false> struct OpPatch {
    static constexpr bool TRANS = false;
    const float* img0; const float* img1;      // stream 0 (imgs), stream 1 (tgts)
    int Bn, Hp, Wp, P, rows;                   // rows = S*B*L
    int kreal;                                 // 3*P*P
    struct Ctx { const float* base; };          // pixel (h*P, w*P) of channel 0, or nullptr
    DEVI void batch(int) {}
    DEVI Ctx ctx(int row) const {
        if (row >= rows) return Ctx{nullptr};
        const int L = Hp * Wp, BL = Bn * L;
        const int s = row / BL, r = row % BL, b = r / L, l = r % L, h = l / Wp, w = l % Wp;
        const float* img = s ? img1 : img0;
        return Ctx{img + ((size_t)b * 3 * Hp * P + h * P) * (size_t)(Wp * P) + w * P};
    }
    DEVI float elem(Ctx c, int k) const {
        if (k >= kreal) return 0.f;
        const int PP = P * P, ch = k / PP, ph = (k % PP) / P, pw = k % P;
        return c.base[((size_t)ch * Hp * P + ph) * (size_t)(Wp * P) + pw];
    }
    DEVI uint4 chunk(Ctx c, int k, int kend) const {
        if (c.base == nullptr || k >= kend) return zero4();
        if constexpr (GEN) {
            if constexpr (sizeof(T) == 2)
                return make_uint4(pack_bf16x2(elem(c, k), elem(c, k + 1)), pack_bf16x2(elem(c, k + 2), elem(c, k + 3)),
                                  pack_bf16x2(elem(c, k + 4), elem(c, k + 5)), pack_bf16x2(elem(c, k + 6), elem(c, k + 7)));
            else
                return make_uint4(__builtin_bit_cast(uint32_t, elem(c, k)), __builtin_bit_cast(uint32_t, elem(c, k + 1)),
                                  __builtin_bit_cast(uint32_t, elem(c, k + 2)), __builtin_bit_cast(uint32_t, elem(c, k + 3)));
        }
        const int PP = P * P, ch = k / PP, ph = (k % PP) / P, pw = k % P;
        const float* src = c.base + ((size_t)ch * Hp * P + ph) * (size_t)(Wp * P) + pw;
        if constexpr (sizeof(T) == 2) {
            const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
            return make_uint4(pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w), pack_bf16x2(b.x, b.y), pack_bf16x2(b.z, b.w));
        } else {
            return *reinterpret_cast<const uint4*>(src);
        }
    }
};
// contraction-major view of the same matrix for the weight gradient: vec(kk = token, r = k index)
template <typename T, bool GEN = false> struct OpPatchT {
    static constexpr bool TRANS = true;
    typedef typename TT<T>::Vec4 Vec4;
    typedef int Ctx;
    const float* img0; const float* img1;
    int Bn, Hp, Wp, P, rows;                   // rows = 3*P*P (the k extent)
    DEVI void batch(int) {}
    DEVI Vec4 vec(int kk, int r, int kend) const {
        Vec4 v; zero_vec(v);
        if (kk >= kend || r >= rows) return v;
        const int L = Hp * Wp, BL = Bn * L;
        const int s = kk / BL, rr = kk % BL, b = rr / L, l = rr % L, h = l / Wp, w = l % Wp;
        const int PP = P * P, ch = r / PP, ph = (r % PP) / P, pw = r % P;
        const float* img = s ? img1 : img0;
        if constexpr (GEN) {                     // 4 consecutive k indices may wrap to the next patch row / channel; r + e >= rows reads 0
            float e4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int re = r + e;
                const int che = re / PP, phe = (re % PP) / P, pwe = re % P;
                e4[e] = re < rows ? img[(((size_t)b * 3 + che) * Hp * P + h * P + phe) * (size_t)(Wp * P) + w * P + pwe] : 0.f;
            }
            return cvt4(e4[0], e4[1], e4[2], e4[3], (T*)nullptr);
        }
        const float4 a = *reinterpret_cast<const float4*>(img + (((size_t)b * 3 + ch) * Hp * P + h * P + ph) * (size_t)(Wp * P) + w * P + pw);
        return cvt4(a.x, a.y, a.z, a.w, (T*)nullptr);
    }
};

// tokens = PE + bias, y stream: masked tokens <- mask_token; + segment token + abs pos (+ type token)
struct EpiPatchTokens {
    float* out; size_t ldo;
    const float* bias; const float* mask_token; const float* seg_x; const float* seg_y; const float* pos;   // pos [L, D]
    const unsigned char* mask; int mask_bstride;             // [B, L] bool (bstride 0 = broadcast one row)
    const float* type_cls; const float* type_ins; const float* seg_type;   // SegGPT (NULL for Painter); seg_type [B]
    int Bn, L, M, N;
    DEVI void operator()(const f32x16 (&acc)[2][2], int ib, int jb, int lane, int) const {
        foreach_acc(acc, ib, jb, lane, [&](int i, int j, float v) {
            if (i < M && j < N) {
                const int BL = Bn * L, s = i / BL, r = i % BL, b = r / L, l = r % L;
                float t = v + bias[j];
                if (s == 1) {
                    const float w = mask[(size_t)b * mask_bstride + l] ? 1.f : 0.f;
                    t = t * (1.f - w) + mask_token[j] * w;
                    t += seg_y[j];
                } else {
                    t += seg_x[j];
                }
                t += pos[(size_t)l * N + j];
                if (seg_type) {
                    const float st = seg_type[b];
                    t += (st == 0.f) ? type_cls[j] : ((st == 1.f) ? type_ins[j] : 0.f);
                }
                out[(size_t)i * ldo + j] = t;
            }
        });
    }
};

template <typename T>
static int patch_fwd_t(const float* imgs, const float* tgts, const T* w, int64_t ldw, EpiPatchTokens ep, int Bn, int Hp, int Wp, int P, int D,
                       hipStream_t st) {
    const int M = 2 * Bn * Hp * Wp, K = 3 * P * P, Kp = (K + 7) / 8 * 8;
    OpN<T> B{w, (size_t)ldw, D, 0};
    if (P % 8 == 0) return launch_gemm<T, 2, 2>(OpPatch<T, false>{imgs, tgts, Bn, Hp, Wp, P, M, K}, B, ep, M, D, K, 1, 1, st);
    return launch_gemm<T, 2, 2>(OpPatch<T, true>{imgs, tgts, Bn, Hp, Wp, P, M, K}, B, ep, M, D, Kp, 1, 1, st);
}
// w [D, 3*P*P] f32 (the conv weight, flattened) -> T [D, Kp], Kp = 3*P*P rounded up to 8, zero padded: the layout pa_patch_embed_fwd
// takes with ldw = Kp (for P % 8 == 0 that is a plain cast)
template <typename T> __global__ void patch_weight_pack_kernel(const float* __restrict__ w, T* __restrict__ out, int D, int K, int Kp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D * Kp) return;
    const int d = i / Kp, k = i - d * Kp;
    out[i] = from_f<T>(k < K ? w[(size_t)d * K + k] : 0.f);
}
extern "C" int pa_patch_weight_pack(int dtype, const float* w, void* out, int D, int P, hipStream_t st) {
    const int K = 3 * P * P, Kp = (K + 7) / 8 * 8, n = D * Kp;
    if (dtype == PA_BF16) PA_LAUNCH(patch_weight_pack_kernel<bf16>, dim3((n + 255) / 256), dim3(256), 0, st, w, (bf16*)out, D, K, Kp);
    else PA_LAUNCH(patch_weight_pack_kernel<float>, dim3((n + 255) / 256), dim3(256), 0, st, w, (float*)out, D, K, Kp);
    LAUNCH_CHECK();
}
extern "C" int pa_patch_embed_fwd(int dtype, const float* imgs, const float* tgts, const void* w, int64_t ldw, const float* bias,
                                  const float* mask_token, const float* seg_x, const float* seg_y, const float* pos,
                                  const unsigned char* mask, int mask_batch_stride, const float* type_cls,
                                  const float* type_ins, const float* seg_type, float* tokens, int batch, int Hp, int Wp, int P,
                                  int D, hipStream_t st) {
    const int Kp = (3 * P * P + 7) / 8 * 8;
    if (P < 1 || ldw < Kp || ldw % 8) return (int)hipErrorInvalidValue;
    EpiPatchTokens ep{tokens, (size_t)D, bias, mask_token, seg_x, seg_y, pos, mask, mask_batch_stride, type_cls, type_ins, seg_type,
                      batch, Hp * Wp, 2 * batch * Hp * Wp, D};
    if (dtype == PA_BF16) return patch_fwd_t<bf16>(imgs, tgts, (const bf16*)w, ldw, ep, batch, Hp, Wp, P, D, st);
    return patch_fwd_t<float>(imgs, tgts, (const float*)w, ldw, ep, batch, Hp, Wp, P, D, st);
}

// ---- bf16 fast path for P % 8 == 0 (every reference factory: P = 16, K = 768): the im2col operand is materialised once (38.5 MB at
// B = 8, an HBM-bound 25 us pass; it is also the X operand of the weight gradient, so it is kept for the backward) and the contraction
// runs on the 256 x 256 LDS-DMA kernel (gemm256.h) instead of the register-staged gather engine: forward 312 -> ~60 us, weight
// gradient 294 -> ~70 us per step.
// cols[t][c*P*P + ph*P + pw] = img_s[b, c, h*P + ph, w*P + pw]   (t = (s, b, h, w) as above)
__global__ __launch_bounds__(256) void patch_im2col_kernel(const float* __restrict__ img0, const float* __restrict__ img1, bf16* __restrict__ cols,
                                                           int Bn, int Hp, int Wp, int P, size_t total8) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;         // one 16-byte chunk = 8 consecutive pw of one patch row
    if (i >= total8) return;
    const int K = 3 * P * P, K8 = K / 8;
    const size_t row = i / K8;
    const int k = (int)(i - row * K8) * 8;
    const int L = Hp * Wp, BL = Bn * L;
    const int s = (int)(row / BL), r = (int)(row % BL), b = r / L, l = r % L, h = l / Wp, w = l % Wp;
    const int PP = P * P, ch = k / PP, ph = (k % PP) / P, pw = k % P;
    const float* src = (s ? img1 : img0) + (((size_t)b * 3 + ch) * Hp * P + h * P + ph) * (size_t)(Wp * P) + w * P + pw;
    const float4 a = *reinterpret_cast<const float4*>(src), c = *reinterpret_cast<const float4*>(src + 4);
    *reinterpret_cast<uint4*>(cols + row * K + k) = make_uint4(pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w), pack_bf16x2(c.x, c.y), pack_bf16x2(c.z, c.w));
}
extern "C" int pa_patch_im2col(const float* imgs, const float* tgts, void* cols, int batch, int Hp, int Wp, int P, hipStream_t st) {
    if (P < 8 || P % 8) return (int)hipErrorInvalidValue;
    const size_t total8 = (size_t)2 * batch * Hp * Wp * (3 * P * P / 8);
    PA_LAUNCH(patch_im2col_kernel, dim3((unsigned)((total8 + 255) / 256)), dim3(256), 0, st, imgs, tgts, (bf16*)cols, batch, Hp, Wp, P, total8);
    LAUNCH_CHECK();
}
// the token-assembly epilogue in the 8-wide form of gemm256.h (same arithmetic as EpiPatchTokens, 8 columns per call)
struct Epi4PatchTokens {
    float* out; size_t ldo;
    const float* bias; const float* mask_token; const float* seg_x; const float* seg_y; const float* pos;
    const unsigned char* mask; int mask_bstride;
    const float* type_cls; const float* type_ins; const float* seg_type;
    int Bn, L, M, N;
    struct Col { float4 a, b; };
    struct Row { int l; float w; int stream; int type; };       // type: 0 = cls token, 1 = ins token, 2 = none
    DEVI static float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
    DEVI Col col(int j) const {
        Col c{make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0)};
        if (j < N) { c.a = ld4(bias + j); c.b = ld4(bias + j + 4); }
        return c;
    }
    DEVI Row row(int i, int) const {
        Row r{0, 0.f, 0, 2};
        if (i < M) {
            const int BL = Bn * L, s = i / BL, rr = i % BL, b = rr / L;
            r.l = rr % L;
            r.stream = s;
            r.w = (s == 1 && mask[(size_t)b * mask_bstride + r.l]) ? 1.f : 0.f;
            if (seg_type) { const float st = seg_type[b]; r.type = st == 0.f ? 0 : (st == 1.f ? 1 : 2); }
        }
        return r;
    }
    DEVI float one(float v, float bj, float mt, float sg, float ps, float tt, const Row& r) const {
        float t = v + bj;
        if (r.stream == 1) { t = t * (1.f - r.w) + mt * r.w; }
        t += sg;
        t += ps;
        if (r.type != 2) t += tt;
        return t;
    }
    DEVI void store(int i, int j, float4 a, float4 b, const Col& c, const Row& r, int) const {
        if (i >= M || j >= N) return;
        const float* sgp = r.stream == 1 ? seg_y : seg_x;
        const float* tp = r.type == 0 ? type_cls : type_ins;
        const float4 m0 = ld4(mask_token + j), m1 = ld4(mask_token + j + 4), s0 = ld4(sgp + j), s1 = ld4(sgp + j + 4);
        const float4 p0 = ld4(pos + (size_t)r.l * N + j), p1 = ld4(pos + (size_t)r.l * N + j + 4);
        float4 t0 = make_float4(0, 0, 0, 0), t1 = t0;
        if (r.type != 2) { t0 = ld4(tp + j); t1 = ld4(tp + j + 4); }
        float* o = out + (size_t)i * ldo + j;
        *reinterpret_cast<float4*>(o) = make_float4(one(a.x, c.a.x, m0.x, s0.x, p0.x, t0.x, r), one(a.y, c.a.y, m0.y, s0.y, p0.y, t0.y, r),
                                                    one(a.z, c.a.z, m0.z, s0.z, p0.z, t0.z, r), one(a.w, c.a.w, m0.w, s0.w, p0.w, t0.w, r));
        *reinterpret_cast<float4*>(o + 4) = make_float4(one(b.x, c.b.x, m1.x, s1.x, p1.x, t1.x, r), one(b.y, c.b.y, m1.y, s1.y, p1.y, t1.y, r),
                                                        one(b.z, c.b.z, m1.z, s1.z, p1.z, t1.z, r), one(b.w, c.b.w, m1.w, s1.w, p1.w, t1.w, r));
    }
};
// tokens from the materialised im2col operand; returns hipErrorInvalidValue where gemm256 does not take the shape (the caller then
// uses pa_patch_embed_fwd)
extern "C" int pa_patch_embed_fwd_cols(const void* cols, const void* w, int64_t ldw, const float* bias, const float* mask_token, const float* seg_x,
                                       const float* seg_y, const float* pos, const unsigned char* mask, int mask_batch_stride,
                                       const float* type_cls, const float* type_ins, const float* seg_type, float* tokens, int batch, int L, int K,
                                       int D, hipStream_t st) {
    const int M = 2 * batch * L;
    if (!g256::ok(M, D, K, false, false, K, ldw) || D % 8) return (int)hipErrorInvalidValue;
    Epi4PatchTokens ep{tokens, (size_t)D, bias, mask_token, seg_x, seg_y, pos, mask, mask_batch_stride, type_cls, type_ins, seg_type, batch, L, M, D};
    return g256::launch<false, false>((const bf16*)cols, (size_t)K, (const bf16*)w, (size_t)ldw, ep, M, D, K, 1, st);
}
extern "C" int pa_patch_cols_ok(int batch, int L, int P, int D) {        // host-only: does the fast path take this shape?
    const int K = 3 * P * P;
    return (P >= 8 && P % 8 == 0 && D % 8 == 0 && g256::ok(2 * batch * L, D, K, false, false, K, K)) ? 1 : 0;
}

struct EpiSlabC {
    float* out; size_t ldo; size_t slab; int M, N;
    DEVI void operator()(const f32x16 (&acc)[2][2], int ib, int jb, int lane, int z) const {
        float* o = out + (size_t)z * slab;
        foreach_acc(acc, ib, jb, lane, [&](int i, int j, float v) {
            if (i < M && j < N) o[(size_t)i * ldo + j] = v;
        });
    }
};
extern "C" int pa_slab_reduce(const float* in, float* out, int64_t n, int nz, int64_t stride, int accumulate, hipStream_t st);

static const int PATCH_WGRAD_SPLITS = 16;
extern "C" int64_t pa_patch_embed_wgrad_workspace_bytes(int D, int P) { return (int64_t)PATCH_WGRAD_SPLITS * D * 3 * P * P * sizeof(float); }
template <typename T>
static int patch_wgrad_t(const T* dpe, const float* imgs, const float* tgts, float* dw, float* ws, int Bn, int Hp, int Wp, int P, int D, hipStream_t st) {
    const int R = 2 * Bn * Hp * Wp, K = 3 * P * P;
    OpT<T> A{dpe, (size_t)D, D, 0};
    int e;
    if (P % 4 == 0) e = launch_gemm<T, 2, 2>(A, OpPatchT<T, false>{imgs, tgts, Bn, Hp, Wp, P, K}, EpiSlabC{ws, (size_t)K, (size_t)D * K, D, K}, D, K, R, PATCH_WGRAD_SPLITS, 1, st);
    else e = launch_gemm<T, 2, 2>(A, OpPatchT<T, true>{imgs, tgts, Bn, Hp, Wp, P, K}, EpiSlabC{ws, (size_t)K, (size_t)D * K, D, K}, D, K, R, PATCH_WGRAD_SPLITS, 1, st);
    if (e) return e;
    return pa_slab_reduce(ws, dw, (int64_t)D * K, PATCH_WGRAD_SPLITS, (int64_t)D * K, 0, st);
}
// dW[D, 3*P*P] = dPE[2BL, D]^T . im2col(imgs;tgts)
extern "C" int pa_patch_embed_wgrad(int dtype, const void* dpe, const float* imgs, const float* tgts, float* dw, void* workspace,
                                    int batch, int Hp, int Wp, int P, int D, hipStream_t st) {
    if (P < 1 || D % 4) return (int)hipErrorInvalidValue;
    if (dtype == PA_BF16) return patch_wgrad_t<bf16>((const bf16*)dpe, imgs, tgts, dw, (float*)workspace, batch, Hp, Wp, P, D, st);
    return patch_wgrad_t<float>((const float*)dpe, imgs, tgts, dw, (float*)workspace, batch, Hp, Wp, P, D, st);
}

// ------------------------------------------------------------------------------- 3x3 conv operands (NHWC, C = 64)
#define CV_C 64
// B(row = pixel, k = tap*64 + cin) = X[b, y+ky-1, x+kx-1, cin]  (zero outside the image)
template <typename T> struct OpConv {
    static constexpr bool TRANS = false;
    const T* x; int Hi, Wi, rows;
    struct Ctx { const T* center; int y, xx; };
    DEVI void batch(int) {}
    DEVI Ctx ctx(int row) const {
        if (row >= rows) return Ctx{nullptr, 0, 0};
        const int hw = Hi * Wi, rem = row % hw, y = rem / Wi, xx = rem % Wi;
        return Ctx{x + (size_t)row * CV_C, y, xx};
    }
    DEVI uint4 chunk(Ctx c, int k, int kend) const {
        if (c.center == nullptr || k >= kend) return zero4();
        const int tap = k / CV_C, cin = k % CV_C, dy = tap / 3 - 1, dx = tap % 3 - 1;
        const int yy = c.y + dy, xq = c.xx + dx;
        if (yy < 0 || yy >= Hi || xq < 0 || xq >= Wi) return zero4();
        return *reinterpret_cast<const uint4*>(c.center + ((ptrdiff_t)dy * Wi + dx) * CV_C + cin);
    }
};
// contraction-major: vec(kk = pixel, r = tap*64 + cin .. +3)
template <typename T> struct OpConvT {
    static constexpr bool TRANS = true;
    typedef typename TT<T>::Vec4 Vec4;
    typedef int Ctx;
    const T* x; int Hi, Wi, rows;     // rows = 9*64
    DEVI void batch(int) {}
    DEVI Vec4 vec(int kk, int r, int kend) const {
        Vec4 v; zero_vec(v);
        if (kk >= kend || r >= rows) return v;
        const int tap = r / CV_C, cin = r % CV_C, dy = tap / 3 - 1, dx = tap % 3 - 1;
        const int hw = Hi * Wi, rem = kk % hw, y = rem / Wi + dy, xq = rem % Wi + dx;
        if (y < 0 || y >= Hi || xq < 0 || xq >= Wi) return v;
        return *reinterpret_cast<const Vec4*>(x + ((ptrdiff_t)kk + (ptrdiff_t)dy * Wi + dx) * CV_C + cin);
    }
};

// Fused tail epilogue.  Orientation: A = conv weights (i = cout, all 64 in the wave tile), B = pixels (j), so a
// lane holds ONE pixel and 32 of its 64 channels (the other 32 live in lane ^ 32): LayerNorm2D and the 1x1 conv
// are in-lane sums plus one half-wave exchange.
template <typename T, bool FAST = false> struct EpiTail {
    const float* b3; const float* gamma; const float* beta; const float* w1; const float* b1;   // w1 [3][64]
    T* y3;           // conv3x3 output + bias, NHWC [pixels][64] (saved for backward; may be NULL)
    float* pred;     // NCHW [B, 3, Hi, Wi]
    int HW, N;       // pixels per image, total pixels
    float eps;
    DEVI void operator()(const f32x16 (&acc)[2][2], int ib, int jb, int lane, int z) const { (*this)(acc, ib, jb, lane, z, nullptr); }
    // stg (tile kernel, bf16): 8 KB of wave-private LDS -- y3 leaves as whole 128-byte pixel rows (c64::px_stage4 / px_write_rows)
    DEVI void operator()(const f32x16 (&acc)[2][2], int ib, int jb, int lane, int, unsigned char* stg) const {
        // ib == 0 (WM == 1): acc[bi][bj][r] -> cout = bi*32 + acc_row(r), pixel = jb + bj*32 + (lane & 31)
        const int g = lane >> 5;
#pragma unroll
        for (int bj = 0; bj < 2; ++bj) {
            const int pix = jb + bj * 32 + (lane & 31);
            float y[32];
            float s = 0.f;
#pragma unroll
            for (int bi = 0; bi < 2; ++bi)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const float4 bb = *reinterpret_cast<const float4*>(b3 + bi * 32 + 8 * rg + 4 * g);
                    const int o = bi * 16 + rg * 4;
                    y[o + 0] = acc[bi][bj][rg * 4 + 0] + bb.x;
                    y[o + 1] = acc[bi][bj][rg * 4 + 1] + bb.y;
                    y[o + 2] = acc[bi][bj][rg * 4 + 2] + bb.z;
                    y[o + 3] = acc[bi][bj][rg * 4 + 3] + bb.w;
                    // round to T first so that forward and backward normalise exactly the same values
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[o + e] = to_f(from_f<T>(y[o + e]));
                    if (y3 && pix < N) {
                        typename TT<T>::Vec4 pk = cvt4(y[o], y[o + 1], y[o + 2], y[o + 3], (T*)nullptr);
                        if constexpr (std::is_same<T, bf16>::value) {
                            if (stg != nullptr) c64::px_stage4(stg + bj * 4096, lane & 31, bi * 32 + 8 * rg + 4 * g, pk);
                            else *reinterpret_cast<typename TT<T>::Vec4*>(y3 + (size_t)pix * CV_C + bi * 32 + 8 * rg + 4 * g) = pk;
                        } else {
                            *reinterpret_cast<typename TT<T>::Vec4*>(y3 + (size_t)pix * CV_C + bi * 32 + 8 * rg + 4 * g) = pk;
                        }
                    }
                    s += (y[o] + y[o + 1]) + (y[o + 2] + y[o + 3]);
                }
            if constexpr (std::is_same<T, bf16>::value) {
                if (y3 && stg != nullptr) {          // (same-wave LDS operations are ordered: no barrier)
                    const int p0 = jb + bj * 32;
                    c64::px_write_rows(stg + bj * 4096, lane, [&](int px) { return p0 + px < N ? y3 + (size_t)(p0 + px) * CV_C : (T*)nullptr; });
                }
            }
            s += lane_xor32(s);
            const float mu = s * (1.f / CV_C);
            float q = 0.f;
#pragma unroll
            for (int e = 0; e < 32; ++e) { const float d = y[e] - mu; q += d * d; }
            q += lane_xor32(q);
            const float rs = 1.f / sqrtf(q * (1.f / CV_C) + eps);
            float o0 = 0.f, o1 = 0.f, o2 = 0.f;
#pragma unroll
            for (int bi = 0; bi < 2; ++bi)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int c0 = bi * 32 + 8 * rg + 4 * g, o = bi * 16 + rg * 4;
                    const float4 ga = *reinterpret_cast<const float4*>(gamma + c0), be = *reinterpret_cast<const float4*>(beta + c0);
                    const float4 wa = *reinterpret_cast<const float4*>(w1 + c0), wb = *reinterpret_cast<const float4*>(w1 + CV_C + c0),
                                 wc = *reinterpret_cast<const float4*>(w1 + 2 * CV_C + c0);
                    const float z0 = (y[o + 0] - mu) * rs * ga.x + be.x, z1 = (y[o + 1] - mu) * rs * ga.y + be.y,
                                z2 = (y[o + 2] - mu) * rs * ga.z + be.z, z3 = (y[o + 3] - mu) * rs * ga.w + be.w;
                    float a0, a1, a2, a3;
                    if constexpr (FAST) {       // packed pairs (round 6): the epilogue is VALU-bound (busy 0.55), half of it this GELU
                        const f32x2_t p01 = gelu_fast2(z0, z1), p23 = gelu_fast2(z2, z3);
                        a0 = p01[0]; a1 = p01[1]; a2 = p23[0]; a3 = p23[1];
                    } else {
                        a0 = gelu_f(z0); a1 = gelu_f(z1); a2 = gelu_f(z2); a3 = gelu_f(z3);
                    }
                    o0 += (a0 * wa.x + a1 * wa.y) + (a2 * wa.z + a3 * wa.w);
                    o1 += (a0 * wb.x + a1 * wb.y) + (a2 * wb.z + a3 * wb.w);
                    o2 += (a0 * wc.x + a1 * wc.y) + (a2 * wc.z + a3 * wc.w);
                }
            o0 += lane_xor32(o0);
            o1 += lane_xor32(o1);
            o2 += lane_xor32(o2);
            if (pix < N && g == 0) {
                const int b = pix / HW, rem = pix % HW;
                float* pp = pred + (size_t)b * 3 * HW + rem;
                pp[0] = o0 + b1[0];
                pp[(size_t)HW] = o1 + b1[1];
                pp[(size_t)2 * HW] = o2 + b1[2];
            }
        }
    }
};

template <typename T>
static int tail_fwd_t(const T* x, const T* w3r, const float* b3, const float* gamma, const float* beta, const float* w1, const float* b1,
                      T* y3, float* pred, int Bn, int Hi, int Wi, float eps, hipStream_t st) {
    const int N = Bn * Hi * Wi;
    if constexpr (std::is_same<T, bf16>::value) {
        if (c64::ok(Bn, Hi, Wi))
            return c64::launch_tile(x, w3r, EpiTail<bf16, true>{b3, gamma, beta, w1, b1, y3, pred, Hi * Wi, N, eps}, Bn, Hi, Wi, st);
    }
    OpN<T> A{w3r, (size_t)9 * CV_C, CV_C, 0};
    OpConv<T> B{x, Hi, Wi, N};
    return launch_gemm<T, 1, 4>(A, B, EpiTail<T>{b3, gamma, beta, w1, b1, y3, pred, Hi * Wi, N, eps}, CV_C, N, 9 * CV_C, 1, 1, st);
}
// x: NHWC [B,Hi,Wi,64] T; w3r: T [64 cout][9 taps][64 cin] (pa_conv3x3_pack); w1: f32 [3,64]; pred: f32 NCHW
extern "C" int pa_decoder_tail_fwd(int dtype, const void* x, const void* w3r, const float* b3, const float* ln_gamma,
                                   const float* ln_beta, const float* w1, const float* b1, void* y3, float* pred, int batch,
                                   int Hi, int Wi, float eps, hipStream_t st) {
    if (dtype == PA_BF16)
        return tail_fwd_t<bf16>((const bf16*)x, (const bf16*)w3r, b3, ln_gamma, ln_beta, w1, b1, (bf16*)y3, pred, batch, Hi, Wi, eps, st);
    return tail_fwd_t<float>((const float*)x, (const float*)w3r, b3, ln_gamma, ln_beta, w1, b1, (float*)y3, pred, batch, Hi, Wi, eps, st);
}

// weight repacks: w3 [cout][cin][3][3] f32 -> w3r [cout][tap][cin] T (forward) and wf [cin][tap'][cout] T with
// tap' = 8 - tap (data gradient = correlation with the flipped kernel)
template <typename T> __global__ void conv3x3_pack_kernel(const float* w3, T* w3r, T* wf) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= CV_C * CV_C * 9) return;
    const int co = i / (CV_C * 9), rem = i % (CV_C * 9), ci = rem / 9, tap = rem % 9;
    const float v = w3[i];
    w3r[(size_t)co * 9 * CV_C + tap * CV_C + ci] = from_f<T>(v);
    wf[(size_t)ci * 9 * CV_C + (8 - tap) * CV_C + co] = from_f<T>(v);
}
extern "C" int pa_conv3x3_pack(int dtype, const float* w3, void* w3r, void* wf, hipStream_t st) {
    const int n = CV_C * CV_C * 9;
    if (dtype == PA_BF16) PA_LAUNCH(conv3x3_pack_kernel<bf16>, dim3((n + 255) / 256), dim3(256), 0, st, w3, (bf16*)w3r, (bf16*)wf);
    else PA_LAUNCH(conv3x3_pack_kernel<float>, dim3((n + 255) / 256), dim3(256), 0, st, w3, (float*)w3r, (float*)wf);
    LAUNCH_CHECK();
}

// data gradient of the 3x3 conv, written straight into the token-major layout of decoder_embed's output:
// dE[(b,h,w)][(p*P+q)*64 + c] = dX[b, h*P+p, w*P+q, c]     (inverse of the pixel shuffle, models_painter.py:424-428)
template <typename T> struct EpiUnshuf {
    T* out; int Hp, Wp, P, M;     // M = pixels
    DEVI void operator()(const f32x16 (&acc)[2][2], int ib, int jb, int lane, int) const {
        const int Wi = Wp * P, HW = Hp * P * Wi;
        foreach_acc(acc, ib, jb, lane, [&](int i, int j, float v) {
            if (i < M && j < CV_C) {
                const int b = i / HW, rem = i % HW, y = rem / Wi, x = rem % Wi;
                const int h = y / P, p = y % P, w = x / P, q = x % P;
                const size_t row = ((size_t)b * Hp + h) * Wp + w;
                out[row * (size_t)(P * P * CV_C) + (size_t)(p * P + q) * CV_C + j] = from_f<T>(v);
            }
        });
    }
};
// the same for the tile kernel's orientation (lane = pixel, registers = channels): 8-byte runs of 4 channels
struct EpiUnshufPx {
    bf16* out; int Hp, Wp, P;
    // stg: 8 KB of wave-private LDS (round 6): a pixel's 64 channels leave as one 128-byte row, 8 lanes x 16 bytes (c64::px_write_rows)
    DEVI void operator()(const f32x16 (&acc)[2][2], int, int jb, int lane, int, unsigned char* stg) const {
        const int Wi = Wp * P, HW = Hp * P * Wi, g = lane >> 5;
#pragma unroll
        for (int bj = 0; bj < 2; ++bj) {
#pragma unroll
            for (int bi = 0; bi < 2; ++bi)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg)
                    c64::px_stage4(stg + bj * 4096, lane & 31, bi * 32 + 8 * rg + 4 * g,
                                   make_uint2(pack_bf16x2(acc[bi][bj][rg * 4], acc[bi][bj][rg * 4 + 1]), pack_bf16x2(acc[bi][bj][rg * 4 + 2], acc[bi][bj][rg * 4 + 3])));
            const int p0 = jb + bj * 32;
            c64::px_write_rows(stg + bj * 4096, lane, [&](int px) {
                const int pix = p0 + px;
                const int b = pix / HW, rem = pix % HW, y = rem / Wi, x = rem % Wi;
                const int h = y / P, p = y % P, w = x / P, q = x % P;
                return out + (((size_t)b * Hp + h) * Wp + w) * (size_t)(P * P * CV_C) + (size_t)(p * P + q) * CV_C;
            });
        }
    }
};
template <typename T>
static int conv_dgrad_t(const T* dy3, const T* wf, T* dE, int Bn, int Hp, int Wp, int P, hipStream_t st) {
    const int Hi = Hp * P, Wi = Wp * P, N = Bn * Hi * Wi;
    if constexpr (std::is_same<T, bf16>::value) {
        if (c64::ok(Bn, Hi, Wi)) return c64::launch_tile(dy3, wf, EpiUnshufPx{dE, Hp, Wp, P}, Bn, Hi, Wi, st);
    }
    OpConv<T> A{dy3, Hi, Wi, N};
    OpN<T> B{wf, (size_t)9 * CV_C, CV_C, 0};
    return launch_gemm<T, 4, 1>(A, B, EpiUnshuf<T>{dE, Hp, Wp, P, N}, N, CV_C, 9 * CV_C, 1, 1, st);
}
extern "C" int pa_conv3x3_dgrad_unshuffle(int dtype, const void* dy3, const void* wf, void* dE, int batch, int Hp, int Wp, int P,
                                          hipStream_t st) {
    if (dtype == PA_BF16) return conv_dgrad_t<bf16>((const bf16*)dy3, (const bf16*)wf, (bf16*)dE, batch, Hp, Wp, P, st);
    return conv_dgrad_t<float>((const float*)dy3, (const float*)wf, (float*)dE, batch, Hp, Wp, P, st);
}

// weight gradient: dW3[cout][cin][tap] = sum_pix dY3[pix][cout] * X[pix + off(tap)][cin]
struct EpiSlabConvW {   // slab layout already in the parameter's [cout][cin][3][3] order
    float* out; size_t slab;
    DEVI void operator()(const f32x16 (&acc)[2][2], int ib, int jb, int lane, int z) const {
        float* o = out + (size_t)z * slab;
        foreach_acc(acc, ib, jb, lane, [&](int i, int j, float v) {
            if (i < CV_C && j < 9 * CV_C) o[(size_t)i * 9 * CV_C + (j % CV_C) * 9 + j / CV_C] = v;
        });
    }
};
static int conv_wgrad_splits(int npix) {
    int s = npix / 8192;
    if (s < 1) s = 1;
    if (s > 192) s = 192;
    return s;
}
extern "C" int64_t pa_conv3x3_wgrad_workspace_bytes(int batch, int Hi, int Wi) {
    int s = conv_wgrad_splits(batch * Hi * Wi);
    if (c64::ok(batch, Hi, Wi)) {
        const int groups = c64::wgrad_groups(batch * (Hi / c64::WTH) * (Wi / c64::TW));
        if (groups > s) s = groups;
    }
    return (int64_t)s * CV_C * CV_C * 9 * sizeof(float);
}
template <typename T>
static int conv_wgrad_t(const T* dy3, const T* x, float* dw, float* ws, int Bn, int Hi, int Wi, hipStream_t st) {
    const int N = Bn * Hi * Wi;
    if constexpr (std::is_same<T, bf16>::value) {
        if (c64::ok(Bn, Hi, Wi)) {
            const int groups = c64::wgrad_groups(Bn * (Hi / c64::WTH) * (Wi / c64::TW));
            int e = c64::launch_wgrad(dy3, x, ws, Bn, Hi, Wi, st);
            if (e) return e;
            return pa_slab_reduce(ws, dw, (int64_t)c64::W_SLAB, groups, (int64_t)c64::W_SLAB, 0, st);
        }
    }
    OpT<T> A{dy3, (size_t)CV_C, CV_C, 0};
    OpConvT<T> B{x, Hi, Wi, 9 * CV_C};
    const int s = conv_wgrad_splits(N);
    int e = launch_gemm<T, 1, 4>(A, B, EpiSlabConvW{ws, (size_t)CV_C * CV_C * 9}, CV_C, 9 * CV_C, N, s, 1, st);
    if (e) return e;
    return pa_slab_reduce(ws, dw, (int64_t)CV_C * CV_C * 9, s, (int64_t)CV_C * CV_C * 9, 0, st);
}
extern "C" int pa_conv3x3_wgrad(int dtype, const void* dy3, const void* x, float* dw, void* workspace, int batch, int Hi, int Wi,
                                hipStream_t st) {
    if (dtype == PA_BF16) return conv_wgrad_t<bf16>((const bf16*)dy3, (const bf16*)x, dw, (float*)workspace, batch, Hi, Wi, st);
    return conv_wgrad_t<float>((const float*)dy3, (const float*)x, dw, (float*)workspace, batch, Hi, Wi, st);
}
