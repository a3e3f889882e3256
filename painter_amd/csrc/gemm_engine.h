// One MFMA contraction engine for every GEMM-shaped op on the path:
//   D[i][j] = sum_k A(i,k) * B(j,k)       i in [0,M), j in [0,N), k in [kb,ke)
// A and B are *operand classes* that say how to fetch 16-byte pieces of their tile, so the same
// main loop serves nn.Linear forward (X, W both K-contiguous), dgrad (W contraction-major),
// wgrad (both contraction-major), the patch-embed im2col gather, the 3x3 conv implicit GEMM and
// the rel-pos gradient contraction.  Epilogue functors receive the wave's 64x64 accumulator tile.
//
// Tile: (WM*64) x (WN*64) per workgroup of WM*WN waves, each wave 2x2 MFMA 32x32 blocks; K tile =
// 128 bytes per row (64 bf16 / 32 f32); LDS double-buffered, register-staged (global -> VGPR ->
// swizzled LDS, next tile's loads in flight under the current tile's MFMAs), one barrier per tile.
#pragma once
#include "common.h"
#include <type_traits>

// ------------------------------------------------------------------------------- operand classes
template <typename T> struct OpN {   // rows x K, K contiguous ("row-major, K innermost")
    static constexpr bool TRANS = false;
    const T* p; size_t ld; int rows; size_t bstride;
    typedef const T* Ctx;          // per-row context, computed once before the K loop
    DEVI void batch(int b) { p += (size_t)b * bstride; }
    DEVI Ctx ctx(int row) const { return row < rows ? p + (size_t)row * ld : nullptr; }
    DEVI uint4 chunk(Ctx c, int k, int kend) const {
        if (c != nullptr && k < kend) return *reinterpret_cast<const uint4*>(c + k);
        return zero4();
    }
};
template <typename T> struct OpT {   // K x rows, rows contiguous (contraction index is the slow one)
    static constexpr bool TRANS = true;
    typedef typename TT<T>::Vec4 Vec4;
    typedef int Ctx;               // unused for contraction-major operands
    const T* p; size_t ld; int rows; size_t bstride;
    DEVI void batch(int b) { p += (size_t)b * bstride; }
    DEVI Vec4 vec(int kk, int r, int kend) const {
        Vec4 v; zero_vec(v);
        if (kk < kend && r < rows) v = *reinterpret_cast<const Vec4*>(p + (size_t)kk * ld + r);
        return v;
    }
};

template <class F> DEVI void foreach_acc(const f32x16 (&acc)[2][2], int ib, int jb, int lane, F f) {
    const int jl = lane & 31;
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
        for (int bj = 0; bj < 2; ++bj)
#pragma unroll
            for (int r = 0; r < 16; ++r) f(ib + bi * 32 + acc_row(r, lane), jb + bj * 32 + jl, acc[bi][bj][r]);
}

// ------------------------------------------------------------------------------- the kernel
template <typename T, int WM, int WN, class AOp, class BOp, class Epi>
__global__ __launch_bounds__(WM * WN * 64) void gemm_kernel(AOp A, BOp B, Epi epi, int M, int N, int K,
                                                             int klen, int nbatch) {
    constexpr int NT = WM * WN * 64, NW = WM * WN;
    constexpr int BM = WM * 64, BN = WN * 64;
    constexpr int BK = TT<T>::BK, EPC = TT<T>::EPC, KS = TT<T>::KSTEPS, SPF = TT<T>::SPF;
    typedef typename TT<T>::Vec4 Vec4;
    constexpr int STAGE = (BM + BN) * 128;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int tiles_n = (N + BN - 1) / BN;
    const int tn = blockIdx.x % tiles_n, tm = blockIdx.x / tiles_n;
    const int i0 = tm * BM, j0 = tn * BN;
    const int z = blockIdx.z, bat = z % nbatch, split = z / nbatch;
    A.batch(bat);
    B.batch(bat);
    const int kb = split * klen;
    const int ke = min(K, kb + klen);
    const int nk = (ke > kb) ? (ke - kb + BK - 1) / BK : 0;

    // staging registers (only the set matching the operand kind survives optimisation)
    constexpr int CA = (BM * 8 + NT - 1) / NT, CB = (BN * 8 + NT - 1) / NT;
    constexpr int PASS_K = BK / 16;
    constexpr int NPA = ((BM / 64) * PASS_K + NW - 1) / NW, NPB = ((BN / 64) * PASS_K + NW - 1) / NW;
    uint4 ra[CA], rb[CB];
    Vec4 ta[NPA][4], tb[NPB][4];
    typename AOp::Ctx ca[CA];
    typename BOp::Ctx cb[CB];
    if constexpr (!AOp::TRANS) {
#pragma unroll
        for (int i = 0; i < CA; ++i) ca[i] = A.ctx(i0 + ((tid + NT * i) >> 3));
    }
    if constexpr (!BOp::TRANS) {
#pragma unroll
        for (int i = 0; i < CB; ++i) cb[i] = B.ctx(j0 + ((tid + NT * i) >> 3));
    }

    // lane -> (row-block, k-block) inside one 64-row x 16-k transposing pass: a 16-lane LDS write
    // group covers 4 row-blocks x 4 k-blocks (bank-conflict free), a wave covers 16 x 4.
    const int t_rb = (lane & 3) + 4 * (lane >> 4);
    const int t_kb = (lane >> 2) & 3;

    auto load_tiles = [&](int k0) {
        if constexpr (!AOp::TRANS) {
#pragma unroll
            for (int i = 0; i < CA; ++i) {
                const int c = tid + NT * i;
                if (BM * 8 % NT == 0 || c < BM * 8) ra[i] = A.chunk(ca[i], k0 + (c & 7) * EPC, ke);
            }
        } else {
#pragma unroll
            for (int it = 0; it < NPA; ++it) {
                const int p = wave + it * NW;
                if (p < (BM / 64) * PASS_K) {
                    const int rbk = (p % (BM / 64)) * 16 + t_rb, kbk = (p / (BM / 64)) * 4 + t_kb;
#pragma unroll
                    for (int i = 0; i < 4; ++i) ta[it][i] = A.vec(k0 + kbk * 4 + i, i0 + rbk * 4, ke);
                }
            }
        }
        if constexpr (!BOp::TRANS) {
#pragma unroll
            for (int i = 0; i < CB; ++i) {
                const int c = tid + NT * i;
                if (BN * 8 % NT == 0 || c < BN * 8) rb[i] = B.chunk(cb[i], k0 + (c & 7) * EPC, ke);
            }
        } else {
#pragma unroll
            for (int it = 0; it < NPB; ++it) {
                const int p = wave + it * NW;
                if (p < (BN / 64) * PASS_K) {
                    const int rbk = (p % (BN / 64)) * 16 + t_rb, kbk = (p / (BN / 64)) * 4 + t_kb;
#pragma unroll
                    for (int i = 0; i < 4; ++i) tb[it][i] = B.vec(k0 + kbk * 4 + i, j0 + rbk * 4, ke);
                }
            }
        }
    };

    auto store_T = [&](unsigned char* base, const Vec4 (&t)[4], int rbk, int kbk) {
        Vec4 o[4];
        transpose4x4(t, o);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = rbk * 4 + j;
            if constexpr (sizeof(T) == 2)
                *reinterpret_cast<Vec4*>(base + lds128(row, kbk >> 1) + (kbk & 1) * 8) = o[j];
            else
                *reinterpret_cast<Vec4*>(base + lds128(row, kbk)) = o[j];
        }
    };

    auto store_tiles = [&](int stage) {
        unsigned char* sa = smem + stage * STAGE;
        unsigned char* sb = sa + BM * 128;
        if constexpr (!AOp::TRANS) {
#pragma unroll
            for (int i = 0; i < CA; ++i) {
                const int c = tid + NT * i;
                if (BM * 8 % NT == 0 || c < BM * 8) *reinterpret_cast<uint4*>(sa + lds128(c >> 3, c & 7)) = ra[i];
            }
        } else {
#pragma unroll
            for (int it = 0; it < NPA; ++it) {
                const int p = wave + it * NW;
                if (p < (BM / 64) * PASS_K)
                    store_T(sa, ta[it], (p % (BM / 64)) * 16 + t_rb, (p / (BM / 64)) * 4 + t_kb);
            }
        }
        if constexpr (!BOp::TRANS) {
#pragma unroll
            for (int i = 0; i < CB; ++i) {
                const int c = tid + NT * i;
                if (BN * 8 % NT == 0 || c < BN * 8) *reinterpret_cast<uint4*>(sb + lds128(c >> 3, c & 7)) = rb[i];
            }
        } else {
#pragma unroll
            for (int it = 0; it < NPB; ++it) {
                const int p = wave + it * NW;
                if (p < (BN / 64) * PASS_K)
                    store_T(sb, tb[it], (p % (BN / 64)) * 16 + t_rb, (p / (BN / 64)) * 4 + t_kb);
            }
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    if (nk > 0) {
        load_tiles(kb);
        store_tiles(0);
    }
    __syncthreads();
    const int fr = lane & 31, fg = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) load_tiles(kb + (kt + 1) * BK);
        const unsigned char* sa = smem + (kt & 1) * STAGE;
        const unsigned char* sb = sa + BM * 128;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            Frag<T> fa[2], fb[2];
            const int slot = (s * 2 + fg) * SPF;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int rowa = wm * 64 + u * 32 + fr, rowb = wn * 64 + u * 32 + fr;
                if constexpr (SPF == 1) {
                    fa[u].set(*reinterpret_cast<const uint4*>(sa + lds128(rowa, slot)));
                    fb[u].set(*reinterpret_cast<const uint4*>(sb + lds128(rowb, slot)));
                } else {
                    fa[u].set(*reinterpret_cast<const uint4*>(sa + lds128(rowa, slot)),
                              *reinterpret_cast<const uint4*>(sa + lds128(rowa, slot + 1)));
                    fb[u].set(*reinterpret_cast<const uint4*>(sb + lds128(rowb, slot)),
                              *reinterpret_cast<const uint4*>(sb + lds128(rowb, slot + 1)));
                }
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) mma(acc[a][b], fa[a], fb[b]);
        }
        if (kt + 1 < nk) store_tiles((kt + 1) & 1);
        __syncthreads();
    }
    epi(acc, i0 + wm * 64, j0 + wn * 64, lane, z);
}

template <typename T, int WM, int WN, class AOp, class BOp, class Epi>
static int launch_gemm(AOp A, BOp B, Epi epi, int M, int N, int K, int nsplit, int nbatch, hipStream_t st) {
    constexpr int BM = WM * 64, BN = WN * 64, BK = TT<T>::BK;
    const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    const int nku = (K + BK - 1) / BK;
    const int klen = ((nku + nsplit - 1) / nsplit) * BK;
    constexpr size_t smem = 2 * (BM + BN) * 128;
    auto kern = gemm_kernel<T, WM, WN, AOp, BOp, Epi>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    if (tiles <= 0 || M <= 0 || N <= 0) return 0;
    PA_LAUNCH(kern, dim3(tiles, 1, nsplit * nbatch), dim3(WM * WN * 64), smem, st, A, B, epi, M, N, K,
                       klen, nbatch);
    return (int)hipGetLastError();
}
