// Micro-benchmarks of the gfx950 facts the attention kernels are designed around (diagnostics; built and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/ubench.hip -o /tmp/ubench && /tmp/ubench).
// Every kernel: one workgroup per CU (grid 256), each wave times its own loop with s_memtime and lane 0 stores the cycle count.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define DEVI __device__ __forceinline__

DEVI f32x16 mfma(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
DEVI void stamp(uint64_t* out, uint64_t t0, int slot) {
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[(size_t)blockIdx.x * 16 + (threadIdx.x >> 6) * 2 + slot] = t1 - t0;
}

// 16 MFMAs per iteration spread over CH independent accumulator chains
template <int CH> DEVI f32x16 mfma_loop(int iters, bf16x8 a, bf16x8 b) {
    f32x16 acc[CH];
    for (int c = 0; c < CH; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k % CH] = mfma(a, b, acc[k % CH]);
    }
    f32x16 s = acc[0];
    for (int c = 1; c < CH; ++c)
        for (int r = 0; r < 16; ++r) s[r] += acc[c][r];
    return s;
}
// 16 exp2 + NF fma per iteration on 16 independent registers
template <int NF> DEVI float valu_loop(int iters, float seed) {
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = seed + i * 0.01f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = __builtin_amdgcn_exp2f(-__builtin_fabsf(x[i]));
#pragma unroll
        for (int k = 0; k < NF; ++k) x[k & 15] = __builtin_fmaf(x[k & 15], 0.999f, 0.001f);
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += x[i];
    return s;
}
DEVI bf16x8 mk(float v) {
    bf16x8 r;
    for (int i = 0; i < 8; ++i) r[i] = (__bf16)(v + 0.001f * i);
    return r;
}

template <int CH> __global__ void k_mfma(uint64_t* out, float* sink, int iters) {
    const bf16x8 a = mk(threadIdx.x * 1e-3f), b = mk(0.5f);
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    f32x16 s = mfma_loop<CH>(iters, a, b);
    stamp(out, t0, 0);
    if (s[3] == 12345.f) sink[0] = s[0];
}
template <int NF> __global__ void k_valu(uint64_t* out, float* sink, int iters) {
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    float s = valu_loop<NF>(iters, threadIdx.x * 1e-3f);
    stamp(out, t0, 0);
    if (s == 12345.f) sink[0] = s;
}
// MODE 0: waves 0-3 MFMA, waves 4-7 VALU (pairs w, w+4); MODE 1: even waves MFMA, odd waves VALU (pairs w, w+1); PRIO: s_setprio(1) on the MFMA waves
template <int MODE, int PRIO, int NF> __global__ void k_pair(uint64_t* out, float* sink, int iters) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool is_m = MODE == 0 ? wave < 4 : (wave & 1) == 0;
    const bf16x8 a = mk(threadIdx.x * 1e-3f), b = mk(0.5f);
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    if (is_m) {
        if (PRIO) __builtin_amdgcn_s_setprio(1);
        f32x16 s = mfma_loop<4>(iters, a, b);
        stamp(out, t0, 0);
        if (s[3] == 12345.f) sink[0] = s[0];
    } else {
        float s = valu_loop<NF>(iters, threadIdx.x * 1e-3f);
        stamp(out, t0, 0);
        if (s == 12345.f) sink[0] = s;
    }
}
// alternating phases separated by barriers, as attn3p.hip: group 0 = M V M V ..., group 1 = V M V M ...; one "tile" = 16 MFMA + (16 exp + NF fma)
template <int NF, int CH> __global__ void k_phased(uint64_t* out, float* sink, int tiles) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave >> 2;
    const bf16x8 a = mk(threadIdx.x * 1e-3f), b = mk(0.5f);
    f32x16 acc[CH];
    for (int c = 0; c < CH; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 1e-3f + i * 0.01f;
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    if (grp) __builtin_amdgcn_s_barrier();
    for (int t = 0; t < tiles; ++t) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k % CH] = mfma(a, b, acc[k % CH]);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = __builtin_amdgcn_exp2f(-__builtin_fabsf(x[i] + acc[0][i] * 1e-30f));
#pragma unroll
        for (int k = 0; k < NF; ++k) x[k & 15] = __builtin_fmaf(x[k & 15], 0.999f, 0.001f);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    }
    if (!grp) __builtin_amdgcn_s_barrier();
    stamp(out, t0, 0);
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += x[i];
    for (int c = 0; c < CH; ++c) s += acc[c][5];
    if (s == 12345.f) sink[0] = s;
}
// One wave overlapping its own matrix and vector work: a tile = 16 MFMA (4 chains) + 16 exp + NF fma where the vector work of tile t
// uses the accumulators of tile t-1 (software pipelined, so the two blocks of one iteration are independent).
// MODE 0: MFMA block, then VALU block (sched_barrier between); MODE 1: sched_group_barrier pattern {1 MFMA, 1 exp, NF/16 VALU} x 16;
// MODE 2: left to the compiler's scheduler.
template <int MODE, int NF> __global__ void k_interleave(uint64_t* out, float* sink, int tiles) {
    const bf16x8 a = mk(threadIdx.x * 1e-3f), b = mk(0.5f);
    f32x16 acc[4], prev;
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    for (int r = 0; r < 16; ++r) prev[r] = threadIdx.x * 1e-3f + r * 0.01f;
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = 0.f;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < tiles; ++t) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k & 3] = mfma(a, b, acc[k & 3]);
        if (MODE == 0) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = __builtin_amdgcn_exp2f(-__builtin_fabsf(prev[i] + x[i] * 1e-3f));
#pragma unroll
        for (int k = 0; k < NF; ++k) x[k & 15] = __builtin_fmaf(x[k & 15], 0.999f, 0.001f);
        if (MODE == 1) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x400, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, NF / 16 + 2, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        prev = acc[0];
    }
    stamp(out, t0, 0);
    float v = 0.f;
    for (int i = 0; i < 16; ++i) v += x[i];
    for (int c = 0; c < 4; ++c) v += acc[c][5];
    if (v == 12345.f) sink[0] = v;
}
__global__ void k_barrier(uint64_t* out, int iters) {
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) __builtin_amdgcn_s_barrier();
    stamp(out, t0, 0);
}

// Packed-fp32 check: every lane evaluates the same recurrence twice -- with v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32
// (inline asm, op_sel broadcast forms as hipcc's SLP vectoriser emits them) and with scalar v_fma_f32 / v_mul_f32 / v_add_f32 -- and
// counts bit differences.  Run alone and beside an MFMA kernel on a second stream.
typedef __attribute__((ext_vector_type(2))) float f32x2;
__global__ void k_pkcheck(unsigned long long* bad, int iters) {
    const float s0 = 1.0f + (threadIdx.x & 63) * 0.001f + blockIdx.x * 1e-5f;
    f32x2 a = {s0, s0 * 0.5f}, acc = {0.f, 0.f};
    float a0 = s0, a1 = s0 * 0.5f, c0 = 0.f, c1 = 0.f;
    const float mu = 0.37f, rs = 1.0003f;
    unsigned nb[4] = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        // vector-typed source: hipcc emits v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 for these (checked in the ISA), as it did for the
        // explicit f32x2 code in the attention kernels; the scalar recurrence below must agree bit for bit
        const f32x2 vmu = {mu, mu}, vrs = {rs, rs}, k9 = {0.999f, 0.999f}, k1 = {0.001f, 0.001f};
        f32x2 t = (a - vmu) * vrs;
        asm volatile("" : "+v"(t));
        acc = __builtin_elementwise_fma(t, a, acc);
        a = __builtin_elementwise_fma(t, k9, k1);
        asm volatile("" : "+v"(a), "+v"(acc));
        const float t0 = (a0 - mu) * rs, t1 = (a1 - mu) * rs;
        c0 = __builtin_fmaf(t0, a0, c0);
        c1 = __builtin_fmaf(t1, a1, c1);
        a0 = __builtin_fmaf(t0, 0.999f, 0.001f);
        a1 = __builtin_fmaf(t1, 0.999f, 0.001f);
        if ((it & 63) == 63) {
            // (element copies first: __builtin_bit_cast applied to a vector-element lvalue reads element 0 with this hipcc)
            const float p0 = acc[0], p1 = acc[1], q0 = a[0], q1 = a[1];
            nb[0] += __float_as_uint(p0) != __float_as_uint(c0);
            nb[1] += __float_as_uint(p1) != __float_as_uint(c1);
            nb[2] += __float_as_uint(q0) != __float_as_uint(a0);
            nb[3] += __float_as_uint(q1) != __float_as_uint(a1);
        }
    }
    for (int i = 0; i < 4; ++i)
        if (nb[i]) atomicAdd(bad + i, (unsigned long long)nb[i]);
}

static uint64_t* d_out;
static float* d_sink;
template <class F> static void run(const char* name, int block, double per, F launch) {
    hipMemset(d_out, 0, 256 * 16 * 8);
    launch();
    hipDeviceSynchronize();
    launch();
    hipDeviceSynchronize();
    std::vector<uint64_t> h(256 * 16);
    hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
    const int nw = block / 64;
    printf("%-58s", name);
    for (int w = 0; w < nw; ++w) {
        std::vector<double> v;
        for (int b = 0; b < 256; ++b) v.push_back((double)h[b * 16 + w * 2]);
        std::sort(v.begin(), v.end());
        printf(" w%d %7.1f", w, v[128] / per);
    }
    printf("   (cycles per unit, median over 256 CUs)\n");
    fflush(stdout);
}

int main() {
    hipMalloc(&d_out, 256 * 16 * 8);
    hipMalloc(&d_sink, 64);
    const int IT = 2000;
#define L(k, blk, ...) [&] { hipLaunchKernelGGL(k, dim3(256), dim3(blk), 0, 0, __VA_ARGS__); }
    printf("s_memtime ticks; unit = one MFMA (mfma rows), one v_exp_f32 (valu rows, NF = extra fma per 16 exp), one tile (phased rows)\n");
    run("mfma 1 chain, 1 wave/SIMD", 256, IT * 16.0, L(k_mfma<1>, 256, d_out, d_sink, IT));
    run("mfma 2 chains, 1 wave/SIMD", 256, IT * 16.0, L(k_mfma<2>, 256, d_out, d_sink, IT));
    run("mfma 4 chains, 1 wave/SIMD", 256, IT * 16.0, L(k_mfma<4>, 256, d_out, d_sink, IT));
    run("mfma 1 chain, 2 waves/SIMD", 512, IT * 16.0, L(k_mfma<1>, 512, d_out, d_sink, IT));
    run("mfma 4 chains, 2 waves/SIMD", 512, IT * 16.0, L(k_mfma<4>, 512, d_out, d_sink, IT));
    run("valu 16 exp, 1 wave/SIMD", 256, IT * 16.0, L(k_valu<0>, 256, d_out, d_sink, IT));
    run("valu 16 exp + 48 fma, 1 wave/SIMD", 256, IT * 16.0, L(k_valu<48>, 256, d_out, d_sink, IT));
    run("valu 16 exp, 2 waves/SIMD", 512, IT * 16.0, L(k_valu<0>, 512, d_out, d_sink, IT));
    run("valu 16 exp + 48 fma, 2 waves/SIMD", 512, IT * 16.0, L(k_valu<48>, 512, d_out, d_sink, IT));
    run("valu 16 exp + 48 fma, 4 waves/SIMD", 1024, IT * 16.0, L(k_valu<48>, 1024, d_out, d_sink, IT));
    run("pair (w,w+4): w0-3 mfma(16/it) | w4-7 valu(16exp+48fma/it)", 512, IT * 16.0, L((k_pair<0, 0, 48>), 512, d_out, d_sink, IT));
    run("pair (w,w+4) + setprio(1) on mfma waves", 512, IT * 16.0, L((k_pair<0, 1, 48>), 512, d_out, d_sink, IT));
    run("pair (w,w+1): even mfma | odd valu", 512, IT * 16.0, L((k_pair<1, 0, 48>), 512, d_out, d_sink, IT));
    run("pair (w,w+4), valu = 16 exp only", 512, IT * 16.0, L((k_pair<0, 0, 0>), 512, d_out, d_sink, IT));
    run("phased M|V barriers, 4 chains, NF=48 (cycles per tile)", 512, IT * 1.0, L((k_phased<48, 4>), 512, d_out, d_sink, IT));
    run("phased M|V barriers, 1 chain, NF=48 (cycles per tile)", 512, IT * 1.0, L((k_phased<48, 1>), 512, d_out, d_sink, IT));
    run("phased M|V barriers, 4 chains, NF=0 (cycles per tile)", 512, IT * 1.0, L((k_phased<0, 4>), 512, d_out, d_sink, IT));
    run("one wave: 16 mfma then 16 exp + 48 fma, 1 wave/SIMD (cycles per tile)", 256, IT * 1.0, L((k_interleave<0, 48>), 256, d_out, d_sink, IT));
    run("one wave: interleaved by sched_group_barrier, 1 wave/SIMD", 256, IT * 1.0, L((k_interleave<1, 48>), 256, d_out, d_sink, IT));
    run("one wave: compiler's own order, 1 wave/SIMD", 256, IT * 1.0, L((k_interleave<2, 48>), 256, d_out, d_sink, IT));
    run("16 mfma then 16 exp + 48 fma, 2 waves/SIMD (cycles per tile)", 512, IT * 1.0, L((k_interleave<0, 48>), 512, d_out, d_sink, IT));
    run("interleaved by sched_group_barrier, 2 waves/SIMD", 512, IT * 1.0, L((k_interleave<1, 48>), 512, d_out, d_sink, IT));
    run("compiler's own order, 2 waves/SIMD", 512, IT * 1.0, L((k_interleave<2, 48>), 512, d_out, d_sink, IT));
    run("interleaved, 16 exp + 96 fma, 1 wave/SIMD", 256, IT * 1.0, L((k_interleave<1, 96>), 256, d_out, d_sink, IT));
    run("interleaved, 16 exp + 96 fma, 2 waves/SIMD", 512, IT * 1.0, L((k_interleave<1, 96>), 512, d_out, d_sink, IT));
    run("sequential, 16 exp + 96 fma, 2 waves/SIMD", 512, IT * 1.0, L((k_interleave<0, 96>), 512, d_out, d_sink, IT));
    run("s_barrier only, 8 waves (cycles per barrier)", 512, IT * 1.0, L(k_barrier, 512, d_out, IT));
    run("s_barrier only, 4 waves (cycles per barrier)", 256, IT * 1.0, L(k_barrier, 256, d_out, IT));
    {   // packed-fp32 arithmetic alone, and beside MFMA workgroups of another kernel (second stream)
        unsigned long long* d_bad;
        hipMalloc(&d_bad, 32);
        hipStream_t s1, s2;
        hipStreamCreate(&s1);
        hipStreamCreate(&s2);
        for (int mode = 0; mode < 2; ++mode) {
            hipMemset(d_bad, 0, 32);
            hipDeviceSynchronize();
            for (int rep = 0; rep < 40; ++rep) {
                if (mode) hipLaunchKernelGGL(k_mfma<4>, dim3(192), dim3(512), 0, s2, d_out, d_sink, 3000);
                for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(k_pkcheck, dim3(1024), dim3(256), 0, s1, d_bad, 4096);
            }
            hipDeviceSynchronize();
            unsigned long long h[4];
            hipMemcpy(h, d_bad, 32, hipMemcpyDeviceToHost);
            printf("packed-fp32 (compiler-emitted v_pk_add/mul/fma_f32) vs scalar fp32 recurrence, %s: bit mismatches acc.lo %llu acc.hi %llu a.lo %llu a.hi %llu "
                   "of %llu checks each\n", mode ? "beside an MFMA kernel on a second stream" : "alone", h[0], h[1], h[2], h[3], 320ull * 262144 * 64);
        }
    }
    return 0;
}
