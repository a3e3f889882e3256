// Shared device helpers for the gfx950 (CDNA4) kernels of the Painter/SegGPT ViT hot path.
// wave = 64 lanes everywhere; MFMA 32x32 tiles; LDS tiles are XOR-swizzled 128-byte rows.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define DEVI __device__ __forceinline__

// ---------------------------------------------------------------------------------------------
// element traits: T is the GEMM operand / activation storage type (bf16 "fast" build, float
// "parity" build).  A 16-byte chunk holds EPC elements; a K tile is always 128 bytes per row.
template <typename T> struct TT;
template <> struct TT<float> {
    static constexpr int EPC = 4;      // elements per 16-B chunk
    static constexpr int BK = 32;      // K elements per LDS tile row (128 B)
    static constexpr int KSTEPS = 2;   // 16-element MFMA k-steps per tile
    static constexpr int SPF = 2;      // 16-B slots per 8-element fragment
    typedef uint4 Vec4;                // 4 elements
};
template <> struct TT<bf16> {
    static constexpr int EPC = 8;
    static constexpr int BK = 64;
    static constexpr int KSTEPS = 4;
    static constexpr int SPF = 1;
    typedef uint2 Vec4;
};

DEVI float to_f(float x) { return x; }
DEVI float to_f(bf16 x) { return (float)x; }
template <typename T> DEVI T from_f(float x);
template <> DEVI float from_f<float>(float x) { return x; }
template <> DEVI bf16 from_f<bf16>(float x) { return (bf16)x; }   // v_cvt_pk_bf16_f32, RNE

DEVI uint32_t pack_bf16x2(float lo, float hi) {
    uint16_t a = __builtin_bit_cast(uint16_t, (bf16)lo);
    uint16_t b = __builtin_bit_cast(uint16_t, (bf16)hi);
    return (uint32_t)a | ((uint32_t)b << 16);
}
DEVI float bf16_lo(uint32_t w) { return __builtin_bit_cast(float, w << 16); }
DEVI float bf16_hi(uint32_t w) { return __builtin_bit_cast(float, w & 0xffff0000u); }

// 8-element MFMA operand fragment: lane l holds row (l & 31), contraction slots (l >> 5, t), t = 0..7.
// Any slot -> contraction-index map is legal as long as the A and the B operand use the same one.
template <typename T> struct Frag;
template <> struct Frag<bf16> {
    bf16x8 v;
    DEVI void set(uint4 a) { v = __builtin_bit_cast(bf16x8, a); }
};
template <> struct Frag<float> {
    float v[8];
    DEVI void set(uint4 a, uint4 b) {
        v[0] = __builtin_bit_cast(float, a.x); v[1] = __builtin_bit_cast(float, a.y);
        v[2] = __builtin_bit_cast(float, a.z); v[3] = __builtin_bit_cast(float, a.w);
        v[4] = __builtin_bit_cast(float, b.x); v[5] = __builtin_bit_cast(float, b.y);
        v[6] = __builtin_bit_cast(float, b.z); v[7] = __builtin_bit_cast(float, b.w);
    }
};

// D[i][j] += sum_slots A[i][slot] * B[j][slot];  D layout (all dtypes): lane holds j = lane & 31,
// i = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5), reg = 0..15.
DEVI void mma(f32x16& c, const Frag<bf16>& a, const Frag<bf16>& b) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, c, 0, 0, 0);
}
DEVI void mma(f32x16& c, const Frag<float>& a, const Frag<float>& b) {
#pragma unroll
    for (int t = 0; t < 8; ++t) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[t], b.v[t], c, 0, 0, 0);
}
DEVI int acc_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// ---------------------------------------------------------------------------------------------
// LDS tile layouts (byte offsets).  Rows are 64/128/256 bytes; the 16-B (or 8-B) unit index is
// XOR-swizzled with row bits so that every ds_read_b128/b64 lane group and every staging
// ds_write hits distinct banks (derivation in DESIGN.md "LDS layouts").
DEVI int lds128(int row, int slot) { return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4); }   // 8 slots of 16 B
DEVI int lds256(int row, int slot) { return row * 256 + ((slot ^ (row & 15)) << 4); }         // 16 slots of 16 B
DEVI int lds64(int row, int gran) { return row * 64 + ((gran ^ ((row >> 2) & 7)) << 3); }     // 8 granules of 8 B

// ---------------------------------------------------------------------------------------------
// Cross-lane exchanges on the VALU only (DPP within a row of 16 lanes, gfx950's v_permlane16_swap / v_permlane32_swap across rows).
// hipcc lowers __shfl_xor to ds_bpermute_b32, i.e. a round trip through the LDS unit per butterfly step; these forms never leave the
// SIMD and cost one VALU op each.  (They were introduced while chasing a rare LayerNorm-backward mismatch under two-stream execution;
// that turned out to be SLP-packed fp32 math, see build.py's -fno-slp-vectorize and DESIGN.md section 6 -- the DPP forms stayed
// because they are cheaper.)
template <int CTRL> DEVI float lane_dpp(float v) {
    const int i = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(i, i, CTRL, 0xf, 0xf, false));
}
DEVI float lane_xor1(float v) { return lane_dpp<0xB1>(v); }     // quad_perm [1,0,3,2]
DEVI float lane_xor2(float v) { return lane_dpp<0x4E>(v); }     // quad_perm [2,3,0,1]
DEVI float lane_xor16(float v) {                                 // value of lane ^ 16
    const uint32_t u = __builtin_bit_cast(uint32_t, v);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return __builtin_bit_cast(float, (threadIdx.x & 16) ? r[0] : r[1]);
}
DEVI float lane_xor32(float v) {                                 // value of lane ^ 32
    const uint32_t u = __builtin_bit_cast(uint32_t, v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, (threadIdx.x & 32) ? r[0] : r[1]);
}
// sum / max over the 4 lanes of a quad, the 16 lanes of a row, the whole wave; every lane gets the result.  After the two quad steps
// all lanes of a quad agree, so the mirror permutations (lane i <-> 7-i in a half row, i <-> 15-i in a row) pair distinct groups.
DEVI float quad_sum(float v) { v += lane_xor1(v); v += lane_xor2(v); return v; }
DEVI float row16_sum(float v) { v = quad_sum(v); v += lane_dpp<0x141>(v); v += lane_dpp<0x140>(v); return v; }
DEVI float wave_sum(float v) { v = row16_sum(v); v += lane_xor16(v); v += lane_xor32(v); return v; }
DEVI float wave_max(float v) {
    v = fmaxf(v, lane_xor1(v)); v = fmaxf(v, lane_xor2(v));
    v = fmaxf(v, lane_dpp<0x141>(v)); v = fmaxf(v, lane_dpp<0x140>(v));
    v = fmaxf(v, lane_xor16(v)); v = fmaxf(v, lane_xor32(v));
    return v;
}

// tuning knob shared across translation units: K splits of the rel-pos table-gradient GEMM (0 = built-in default; pa_debug_set(6, n)).
// The engine asks for 4 when that GEMM runs on the side stream beside the data-gradient chain (fewer, longer workgroups: 54.54 -> 54.35
// ms/step), the stand-alone optimum is 16.
inline int g_relpos_splits = 0;
// pa_debug_set(7, v): 0 = default (fused rel-pos table gradient in the generation-3 dQ kernel unless PA_ATTN3_FUSE_RELPOS=0), 1 = off, 2 = on
inline int g_attn3_fuse = 0;
// pa_debug_set(8, v): 0 = default (light attention workgroups NOT dispatched last unless PA_ATTN_LIGHT_LAST=1; round 5), 1 = off, 2 = on
inline int g_attn_light_last = 0;

// pa_debug_set(10, v) (round 5 shared index 5 with gemm256's ILV schedule override): LayerNorm backward variant: 0 = default (rows split over the 4 waves of a workgroup wherever D >= 1024), 1 = one wave per
// row everywhere (the kernel of rounds 1 - 4; still what narrow D runs)
inline int g_ln_bwd_variant = 0;

// pa_debug_set(9, n): tests only -- cap on the number of workgroups of the conv3x3 weight-gradient kernel (0 = the product's 512): with a small
// cap every workgroup walks many tiles, through both ring phases and across column-strip boundaries, at test sizes
inline int g_conv_wgrad_groups = 0;
// pa_debug_set(11 .. 15, v): round-6 experiment knobs, read where they are named: [0] = 11 gemm256 tile patch per XCD (0 = default,
// TR * 16 + TC otherwise), [1] = 12 mixed 224-row + 128-row tiles for the multi-round GEMMs (0 = default on, 1 = off), [2] = 13, [3] = 14, [4] = 15 free
inline int g_misc_knob[5] = {0, 0, 0, 0, 0};

// host-side launch counters of the attention entry points, by kernel family: [0..2] pa_attn_fwd on the generic (attn_fwd.hip) /
// generation-2 (attn2.hip) / generation-3 (attn3.hip) kernels, [3..5] pa_attn_bwd likewise (pa_attn_launch_counts; the model-level tests
// assert with them WHICH kernels a configuration ran on)
inline long long g_attn_counts[6] = {0, 0, 0, 0, 0, 0};

// exact (erf) GELU and its derivative -- nn.GELU default (Painter/models_painter.py:253)
DEVI float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
DEVI float gelu_grad_f(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
    const float pdf = 0.39894228040143268f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// bf16-build variants: erf by Abramowitz-Stegun 7.1.26 (|abs err| < 1.5e-7, far below bf16 resolution) -- two
// transcendentals (rcp, exp2) instead of libm's branchy erff; exp(-x^2/2) is shared between the cdf and the pdf.
DEVI void gelu_parts(float x, float& cdf, float& pdf_unnorm) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    const float e = __builtin_amdgcn_exp2f(-0.72134752044448170f * x * x);          // exp(-x^2 / 2)
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float half_erfc = 0.5f * p * t * e;                                       // 0.5 * erfc(|x| / sqrt 2)
    cdf = x >= 0.f ? 1.0f - half_erfc : half_erfc;
    pdf_unnorm = e;
}
DEVI float gelu_fast(float x) {
    float c, e;
    gelu_parts(x, c, e);
    return x * c;
}
// Two elements at once for the gemm256 epilogues, where the GELU is 33 us of the 154 us fc1 launch with the matrix pipe idle
// (tools/fc1_epilogue.py): the same A-S polynomial on float2 values -- hipcc emits v_pk_mul / v_pk_fma / v_pk_add_f32 for these, half
// the issue slots of the scalar form -- with the constants folded (0.5 into the coefficients, 1/sqrt2 into the rcp argument) and the
// sign handled by copysign instead of compare + select.  Packed fp32 arithmetic is bit-identical to scalar, alone and beside MFMA
// kernels (tools/ubench); only rcp and exp2 stay per element.  Results equal gelu_fast (and the gelu' built from gelu_parts) up to the
// association of the folded constants (1-2 ulp of fp32, far below the bf16 rounding that follows).
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
DEVI void gelu_parts2(f32x2_t x, f32x2_t& cdf, f32x2_t& e) {
    const f32x2_t ax = {fabsf(x[0]), fabsf(x[1])};
    const f32x2_t d = __builtin_elementwise_fma(ax, (f32x2_t){0.23164189f, 0.23164189f}, (f32x2_t){1.0f, 1.0f});      // 1 + 0.3275911 |x| / sqrt 2
    const f32x2_t t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    const f32x2_t xx = x * x * (f32x2_t){-0.72134752044448170f, -0.72134752044448170f};
    e = (f32x2_t){__builtin_amdgcn_exp2f(xx[0]), __builtin_amdgcn_exp2f(xx[1])};                                     // exp(-x^2 / 2)
    f32x2_t p = __builtin_elementwise_fma(t, (f32x2_t){0.5307027145f, 0.5307027145f}, (f32x2_t){-0.7265760135f, -0.7265760135f});   // 0.5 * A-S 7.1.26
    p = __builtin_elementwise_fma(p, t, (f32x2_t){0.7107068705f, 0.7107068705f});
    p = __builtin_elementwise_fma(p, t, (f32x2_t){-0.142248368f, -0.142248368f});
    p = __builtin_elementwise_fma(p, t, (f32x2_t){0.127414796f, 0.127414796f});
    const f32x2_t h = p * t * e;                                                                                  // 0.5 * erfc(|x| / sqrt 2)
    const f32x2_t m = (f32x2_t){0.5f, 0.5f} - h;                                                                  // >= 0
    cdf = (f32x2_t){__builtin_copysignf(m[0], x[0]), __builtin_copysignf(m[1], x[1])} + (f32x2_t){0.5f, 0.5f};
}
DEVI f32x2_t gelu_fast2(float x0, float x1) {
    const f32x2_t x = {x0, x1};
    f32x2_t c, e;
    gelu_parts2(x, c, e);
    return x * c;
}
// gelu(x) and gelu'(x) from one evaluation of the shared parts: the fc1 forward epilogue stores both (act and, for the backward, gelu')
DEVI void gelu_both2(float x0, float x1, f32x2_t& y, f32x2_t& dy) {
    const f32x2_t x = {x0, x1};
    f32x2_t c, e;
    gelu_parts2(x, c, e);
    y = x * c;
    dy = __builtin_elementwise_fma(x * (f32x2_t){0.39894228040143268f, 0.39894228040143268f}, e, c);
}
// 4x4 transpose of a register block: in[i] = 4 consecutive elements (along r) of contraction row i;
// out[j] = the 4 contraction values of element r+j.
DEVI void transpose4x4(const uint4 (&in)[4], uint4 (&out)[4]) {   // float
    out[0] = make_uint4(in[0].x, in[1].x, in[2].x, in[3].x);
    out[1] = make_uint4(in[0].y, in[1].y, in[2].y, in[3].y);
    out[2] = make_uint4(in[0].z, in[1].z, in[2].z, in[3].z);
    out[3] = make_uint4(in[0].w, in[1].w, in[2].w, in[3].w);
}
DEVI void transpose4x4(const uint2 (&in)[4], uint2 (&out)[4]) {   // bf16 (2 per dword, low half first)
    out[0] = make_uint2((in[0].x & 0xffffu) | (in[1].x << 16), (in[2].x & 0xffffu) | (in[3].x << 16));
    out[1] = make_uint2((in[0].x >> 16) | (in[1].x & 0xffff0000u), (in[2].x >> 16) | (in[3].x & 0xffff0000u));
    out[2] = make_uint2((in[0].y & 0xffffu) | (in[1].y << 16), (in[2].y & 0xffffu) | (in[3].y << 16));
    out[3] = make_uint2((in[0].y >> 16) | (in[1].y & 0xffff0000u), (in[2].y >> 16) | (in[3].y & 0xffff0000u));
}

DEVI uint4 zero4() { return make_uint4(0, 0, 0, 0); }
DEVI void zero_vec(uint4& v) { v = make_uint4(0, 0, 0, 0); }
DEVI void zero_vec(uint2& v) { v = make_uint2(0, 0); }

// convert 4 floats to a Vec4 of T
DEVI uint4 cvt4(float a, float b, float c, float d, float*) {
    return make_uint4(__builtin_bit_cast(uint32_t, a), __builtin_bit_cast(uint32_t, b),
                      __builtin_bit_cast(uint32_t, c), __builtin_bit_cast(uint32_t, d));
}
DEVI uint2 cvt4(float a, float b, float c, float d, bf16*) { return make_uint2(pack_bf16x2(a, b), pack_bf16x2(c, d)); }

// hipGetLastError() is sticky across unrelated runtime calls made by the host framework: clear it before launching so the
// value returned after the launch belongs to this launch.
#define PA_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)
#define LAUNCH_CHECK() return (int)hipGetLastError()
