// bf16 MFMA GEMM, second tile shape: 128 x 256 output tile per workgroup of FOUR waves (1 x 4 along j, 128 x 64 per wave), contraction
// tiles of 32 in a ring of three LDS stages (72 KB), so that TWO workgroups are resident per CU (2 waves per SIMD, 256 registers each).
//     D[i][j] = sum_k A(i,k) * B(j,k),   fp32 accumulate,   both operands contraction-contiguous ("K-major": y = x . W^T)
//
// Why a second shape.  gemm256.h (256 x 256, 8 waves, one workgroup per CU) sustains ~1150-1550 TFLOP/s while a tile is in flight, but
// a K = 1024 tile is only 28 us of main loop and its epilogue (LDS transpose, bias / GELU / residual, 128-512 KB of stores that must
// land before the CU takes the next workgroup) adds 11-22 us during which the matrix pipe idles: the forward GEMMs of a block run at
// 0.23-0.32 of peak.  With two resident workgroups one's epilogue and prologue run under the other's main loop, and the tail of a
// launch is made of half-size tiles.  Same fused epilogues (the functors of gemm.hip), same LDS-DMA staging (global_load_lds_dwordx4,
// never through VGPRs), same lane-owns-an-output-row accumulator layout as gemm256.h.
//
// Stage image: A [128 rows][64 B] then B [256 rows][64 B]; the 16-byte chunk c (= 8 k) of row r is stored at chunk c ^ ((r >> 2) & 3):
// every ds_read_b128 row fragment (lane = row, 2 chunks per 16-k step) is conflict-free in each of the instruction's four 16-lane groups.
// Schedule per contraction tile t (stage t % 3):   12 fragment reads | 6 LDS-DMAs of tile t+2 | 16 MFMA | vmcnt(6) | barrier.
// The DMA of tile t+2 overwrites the stage read in iteration t-1 (every wave is past that iteration's barrier); tile t+1 is complete
// for every wave at the barrier that ends iteration t.
#pragma once
#include "gemm256.h"

namespace g128 {

using g256::dma16;
using g256::lds_read128;
using g256::sgpr_ptr;
using g256::wait_vm;

constexpr int BM = 128, BN = 256, BK = 32, NT = 256, RING = 3;
constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
constexpr int LDS_BYTES = RING * STAGE;                        // 73728: two workgroups per CU

template <int N> DEVI void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }

// byte offset (from the operand's tile-0 base) of the 16-byte chunk lane `tid` moves with DMA j
DEVI uint32_t src_off(int tid, int j, int tile0, int rows, uint32_t ld) {
    const int cidx = j * NT + tid;
    const int u = cidx >> 2, p = cidx & 3;
    const int kc = p ^ ((u >> 2) & 3);
    const int row = min(tile0 + u, rows - 1);
    return ((uint32_t)row * ld + kc * 8) * 2u;
}

template <class Epi>
__global__ __launch_bounds__(NT, 2) void gemm128_kernel(const bf16* __restrict__ Ag, uint32_t lda, const bf16* __restrict__ Bg, uint32_t ldb,
                                                        Epi epi, int M, int N, int ktiles, int tiles_n, int stagger) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // The two workgroups a CU holds start together and do identical work, so they would reach their epilogues together, round after
    // round.  The second resident of each CU (workgroups 32..63 of an XCD's run) starts `stagger` shader cycles late; later rounds inherit
    // the phase of the slot they get.
    if (stagger > 0 && (blockIdx.x >> 3) >= 32 && (blockIdx.x >> 3) < 64) {
        const uint64_t t0 = __builtin_amdgcn_s_memtime();
        while ((int64_t)(__builtin_amdgcn_s_memtime() - t0) < (int64_t)stagger) __builtin_amdgcn_s_sleep(16);
    }

    // XCD-aware tile order, blocked 4 x 8 inside an XCD's run (gemm256.h)
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xq = nwg >> 3, xr = nwg & 7, xcd = bid & 7;
    const int tile = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
    int tm, tn;
    {
        constexpr int TR = 8, TC = 8;
        const int tiles_m = (M + BM - 1) / BM;
        const int per_group = TR * tiles_n;
        const int gm = tile / per_group, rem = tile - gm * per_group;
        const int rg = min(TR, tiles_m - gm * TR);
        const int full = tiles_n / TC;
        int cb = rem / (rg * TC), r2 = rem - cb * (rg * TC), cw = TC;
        if (cb >= full) {
            cb = full;
            r2 = rem - full * (rg * TC);
            cw = tiles_n - full * TC;
        }
        const int dm = r2 / cw;
        tm = gm * TR + dm;
        tn = cb * TC + (r2 - dm * cw);
    }
    const int i0 = tm * BM, j0 = tn * BN;
    const int nt = ktiles;

    uint32_t oa[2], ob[4];
#pragma unroll
    for (int j = 0; j < 2; ++j) oa[j] = src_off(tid, j, i0, M, lda);
#pragma unroll
    for (int j = 0; j < 4; ++j) ob[j] = src_off(tid, j, j0, N, ldb);
    const unsigned char* abase = reinterpret_cast<const unsigned char*>(Ag);
    const unsigned char* bbase = reinterpret_cast<const unsigned char*>(Bg);
    unsigned char* const dma_dst = smem + wv * 1024;

    auto stage_tile = [&](int stage, int kt) {
        const unsigned char* sa = sgpr_ptr(abase + (size_t)kt * (BK * 2));
        const unsigned char* sb = sgpr_ptr(bbase + (size_t)kt * (BK * 2));
        unsigned char* d = dma_dst + stage * STAGE;
        dma16(sa + oa[0], d);
        dma16(sa + oa[1], d + 4096);
        dma16(sb + ob[0], d + A_BYTES);
        dma16(sb + ob[1], d + A_BYTES + 4096);
        dma16(sb + ob[2], d + A_BYTES + 8192);
        dma16(sb + ob[3], d + A_BYTES + 12288);
    };

    // fragment read addresses (per lane): row lr of a 32-row block, chunk 2 ks + g, swizzled
    uint32_t ra[2], rb[2];
    {
        const int lr = lane & 31, g = lane >> 5, f = (lr >> 2) & 3;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            ra[ks] = (uint32_t)(lr * 64 + (((2 * ks + g) ^ f) << 4));
            rb[ks] = (uint32_t)(A_BYTES + (wv * 64 + lr) * 64 + (((2 * ks + g) ^ f) << 4));
        }
    }

    f32x16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    auto tile_body = [&](auto stage_c, int T) {
        constexpr int S = decltype(stage_c)::value;
        constexpr int SN = (S + 2) % RING;
        uint4 fa[2][4], fb[2][2];
        // k-step 0 first, so that its MFMAs can start while k-step 1's fragments are still travelling
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            lds_read128<S * STAGE + 0 * 2048>(fa[ks][0], ra[ks]);
            lds_read128<S * STAGE + 0 * 2048>(fb[ks][0], rb[ks]);
            lds_read128<S * STAGE + 1 * 2048>(fa[ks][1], ra[ks]);
            lds_read128<S * STAGE + 1 * 2048>(fb[ks][1], rb[ks]);
            lds_read128<S * STAGE + 2 * 2048>(fa[ks][2], ra[ks]);
            lds_read128<S * STAGE + 3 * 2048>(fa[ks][3], ra[ks]);
        }
        stage_tile(SN, min(T + 2, nt - 1));
        __builtin_amdgcn_sched_barrier(0);
        wait_lgkm<6>();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fb[0][b]), __builtin_bit_cast(bf16x8, fa[0][a]), acc[a][b], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        wait_lgkm<0>();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fb[1][b]), __builtin_bit_cast(bf16x8, fa[1][a]), acc[a][b], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        wait_vm<6>();                       // tile T+1 (issued one iteration ago) has landed; tile T+2's six DMAs may still fly
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;

    stage_tile(0, 0);
    stage_tile(1, min(1, nt - 1));
    wait_vm<6>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    for (int T = 0; T < nt; T += 3) {
        tile_body(I0{}, T);
        if (T + 1 < nt) tile_body(I1{}, T + 1);
        if (T + 2 < nt) tile_body(I2{}, T + 2);
    }
    wait_vm<0>();                           // the clamped tail DMAs must have landed before the stages become the epilogue scratch
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    // ---- epilogue: as gemm256.h (per-wave [32][68]-float LDS scratch, 8 columns per lane, whole row segments per store)
    {
        float* stg = reinterpret_cast<float*>(smem + wv * 8704);
        const int lr = lane & 31, g = lane >> 5, rrow = lane >> 3, c0 = (lane & 7) * 8;
        const int jcol = j0 + wv * 64 + c0;
        const typename Epi::Col col = epi.col(jcol);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            typename Epi::Row rows[2][4];
#pragma unroll
            for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
                for (int st = 0; st < 4; ++st) rows[m2][st] = epi.row(i0 + (half * 2 + m2) * 32 + st * 8 + rrow, jcol);
#pragma unroll
            for (int m2 = 0; m2 < 2; ++m2) {
                const int mb = half * 2 + m2;
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x16& c = acc[mb][nb];
                        *reinterpret_cast<float4*>(stg + lr * 68 + nb * 32 + q * 8 + g * 4) = make_float4(c[q * 4 + 0], c[q * 4 + 1], c[q * 4 + 2], c[q * 4 + 3]);
                    }
                const int ib = i0 + mb * 32;
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    const int r = st * 8 + rrow;
                    const float4 lo = *reinterpret_cast<const float4*>(stg + r * 68 + c0);
                    const float4 hi = *reinterpret_cast<const float4*>(stg + r * 68 + c0 + 4);
                    epi.store(ib + r, jcol, lo, hi, col, rows[m2][st], 0);
                }
            }
        }
    }
}

template <class Epi>
static int launch(const bf16* A, size_t lda, const bf16* B, size_t ldb, Epi epi, int M, int N, int K, hipStream_t st) {
    auto kern = gemm128_kernel<Epi>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    PA_LAUNCH(kern, dim3(tiles_m * tiles_n), dim3(NT), LDS_BYTES, st, A, (uint32_t)lda, B, (uint32_t)ldb, epi, M, N, K / BK, tiles_n, g256::g_dbg[5]);
    return (int)hipGetLastError();
}
static inline bool ok(int M, int N, int K, size_t lda, size_t ldb) {
    if (K % BK || K < 2 * BK || M < 1 || N < 8 || N % 8) return false;
    if ((lda % 8) || (ldb % 8)) return false;
    return (size_t)M * lda * 2 < (1ull << 32) && (size_t)N * ldb * 2 < (1ull << 32);
}

}   // namespace g128
