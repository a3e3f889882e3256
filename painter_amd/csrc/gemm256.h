// bf16 MFMA GEMM for the big nn.Linear contractions of the ViT blocks:
//     D[i][j] = sum_k A(i,k) * B(j,k),   fp32 accumulate
// 256 x 256 output tile per workgroup of 8 waves (2 along i x 4 along j, 128 x 64 per wave, 32x32x16 MFMA), contraction
// tiles of 64, operands staged HBM -> LDS by LDS-DMA (global_load_lds_dwordx4), never through VGPRs.
//
// Each operand may be stored contraction-contiguous ("K-major": X and W of y = x.W^T) or row-contiguous ("M-major":
// W in dX = dY.W, both dY and X in dW = dY^T.X).  K-major tiles are read back as ds_read_b128 fragments from an
// XOR-swizzled [row][64 k] image; M-major tiles keep the memory order ([k][row]) in LDS and are read with the gfx950
// transposing read ds_read_b64_tr_b16, so no transposed copy of an activation or a weight ever exists in HBM.
//
// Schedule (per contraction tile: 4 phases = the 4 quadrants of the wave's 128 x 64 tile):
//     phase = { ds_read this quadrant's new fragments | s_waitcnt vmcnt(6) } barrier
//             { 8 MFMA with the two LDS-DMA pieces of one 16 KB staging unit issued between them (ILV = 2, below) } barrier
// The two wave rows run half a phase apart (the lower row takes one extra barrier on entry), so on every SIMD one wave
// feeds the matrix pipe while its partner issues LDS reads and DMA.  LDS holds 8 staging units (2 stages x {a0,a1,b0,b1},
// a unit = the rows every wave needs in the same phase); a unit is re-staged two phases after its last read and read five
// phases after its DMA was issued, so HBM/L2 latency is covered by ~2.5k cycles of MFMA work and vmcnt never drains to 0
// inside the loop.  Hazard rules followed (MI355X guide, 8-phase template): a DMA is waited for (counted vmcnt) before
// the first barrier of phase p and read in phase p+1 or later; a unit is re-staged >= 2 phases after its last ds_read.
#pragma once
#include "common.h"
#include <cstdlib>
#include <type_traits>

namespace g256 {

constexpr int BM = 256, BN = 256, BK = 64, NT = 512;
constexpr int LDS_BYTES = 131072;
// diagnostics (pa_debug_set): [0] first-round de-phasing in shader cycles, [1] drop epilogue stores, [2] 1 = plain row-major tile order, 2 = blocked order with split = blockIdx.y (the pre-round-5 split-K assignment), [3] wgrad workgroup target
inline int g_dbg[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // [5] (G256_ILV_AB builds) 1 + ILV schedule override
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_cvoid;

// LDS map: A units at [0, 64K): a_sub * 32K + stage * 16K; B units at [64K, 128K) likewise.  Keeps every ds_read
// immediate offset below 64K relative to one per-lane base register per operand.
DEVI constexpr int unit_off(bool isB, int sub, int stage) { return (isB ? 65536 : 0) + sub * 32768 + stage * 16384; }

template <int N> DEVI void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
DEVI void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// source byte offset (from the operand's tile-0 base) of the 16-byte chunk that lane `tid` moves with DMA `j` of a unit
template <bool MM, bool IS_A>
DEVI uint32_t src_off(int tid, int j, int sub, int tile0, int rows, uint32_t ld) {
    const int cidx = j * NT + tid;
    if constexpr (!MM) {
        const int u = cidx >> 3, p = cidx & 7;
        const int kc = p ^ ((u >> 1) & 7);
        int row = IS_A ? ((u >> 6) * 128 + sub * 64 + (u & 63)) : ((u >> 5) * 64 + sub * 32 + (u & 31));
        row = min(tile0 + row, rows - 1);
        return ((uint32_t)row * ld + kc * 8) * 2u;
    } else {
        const int k = cidx >> 4, cp = cidx & 15;
        const int c = cp ^ (4 * (k & 3));
        int n = IS_A ? ((c >> 3) * 128 + sub * 64 + (c & 7) * 8) : ((c >> 2) * 64 + sub * 32 + (c & 3) * 8);
        n = min(tile0 + n, rows - 8);
        return ((uint32_t)k * ld + n) * 2u;
    }
}

// keeps a wave-uniform pointer in SGPRs so that the DMA uses the (sgpr base + 32-bit vgpr offset) addressing form
DEVI const unsigned char* sgpr_ptr(const unsigned char* p) {
    asm volatile("" : "+s"(p));
    return p;
}
DEVI void dma16(const unsigned char* g, unsigned char* l) {
    __builtin_amdgcn_global_load_lds((gbl_cvoid*)g, (lds_void*)l, 16, 0, 0);
}

// fragment reads are inline asm on purpose: a compiler-visible LDS load makes hipcc wait vmcnt(0) for every LDS-DMA in
// flight (it cannot prove the DMA does not alias), which would serialise the pipeline.
template <int OFF> DEVI void lds_read128(uint4& d, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}
template <int OFF> DEVI void lds_read64_tr(uint2& d, uint32_t addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}

// one operand sub-tile of 32 rows x 64 k held by a wave: 4 k-steps of 8 contraction values per lane.  K-major: one
// ds_read_b128 per k-step; M-major: two transposing 64-bit reads (k 0-3 | k 4-7 of the lane's 8) per k-step.
template <bool MM> struct Frag4;
template <> struct Frag4<false> {
    uint4 q0, q1, q2, q3;
    template <int IMM> DEVI void read(const uint32_t (&base)[4], int) {
        lds_read128<IMM>(q0, base[0]);
        lds_read128<IMM>(q1, base[1]);
        lds_read128<IMM>(q2, base[2]);
        lds_read128<IMM>(q3, base[3]);
    }
    template <int KS> DEVI bf16x8 k() const {
        return __builtin_bit_cast(bf16x8, KS == 0 ? q0 : KS == 1 ? q1 : KS == 2 ? q2 : q3);
    }
};
template <> struct Frag4<true> {
    uint2 l0, h0, l1, h1, l2, h2, l3, h3;
    template <int IMM> DEVI void read(const uint32_t (&base)[4], int which_mb) {
        const uint32_t b = base[which_mb];
        lds_read64_tr<IMM + 0 * 4096>(l0, b);
        lds_read64_tr<IMM + 0 * 4096 + 1024>(h0, b);
        lds_read64_tr<IMM + 1 * 4096>(l1, b);
        lds_read64_tr<IMM + 1 * 4096 + 1024>(h1, b);
        lds_read64_tr<IMM + 2 * 4096>(l2, b);
        lds_read64_tr<IMM + 2 * 4096 + 1024>(h2, b);
        lds_read64_tr<IMM + 3 * 4096>(l3, b);
        lds_read64_tr<IMM + 3 * 4096 + 1024>(h3, b);
    }
    template <int KS> DEVI bf16x8 k() const {
        typedef __attribute__((ext_vector_type(2))) uint32_t u2;
        typedef __attribute__((ext_vector_type(4))) uint32_t u4;
        const uint2 l = KS == 0 ? l0 : KS == 1 ? l1 : KS == 2 ? l2 : l3;
        const uint2 h = KS == 0 ? h0 : KS == 1 ? h1 : KS == 2 ? h2 : h3;
        u2 lv = {l.x, l.y}, hv = {h.x, h.y};
        return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lv, hv, 0, 1, 2, 3));
    }
};

// Epilogue contract (j a multiple of 8; N % 8 == 0 is required by ok(); the functor bounds-checks):
//   Col col(j)                      per-lane column constants (bias[j..j+7]), loaded once per tile
//   Row row(i, j)                   per-(row, 8 columns) global inputs (residual, GELU pre-activation), loaded ahead of the stores
//   store(i, j, lo, hi, col, row, split)   lo = D[i][j..j+3], hi = D[i][j+4..j+7]
// Optional column sums of the stored tile (epi_colsum<Epi>::value): the epilogue then calls store_cs(..., cs) instead of store(...), which
// also adds the 8 values AS STORED to the lane's running column sums; after the tile the sums of the lanes that share columns are
// combined (DPP / permlane, fixed order) and colsum_out(part_row, j, sums) writes one partial row per (row tile, wave row) -- the bias
// gradient of the layer whose dY this GEMM's output is, without a separate pass over it (fc1: the fc2 data-gradient GEMM emits dpre).
template <class Epi> struct epi_colsum { static constexpr bool value = false; };
// Which 8 columns a lane handles in the epilogue.  Default (HI_OFF absent = 4): 8 consecutive columns -- one 16-byte store per lane for bf16
// outputs, eight lanes = one 128-byte row segment per instruction.  For fp32 outputs 8 consecutive columns are TWO 16-byte accesses per lane,
// and each of them touches every other 16 bytes of the row segment: two instructions that each half-fill the same cache lines.  A functor
// that declares `static constexpr int HI_OFF = 32` gets columns j..j+3 (lo) and j+32..j+35 (hi) instead: every load / store instruction of
// the fp32 residual read-modify-write and of the weight-gradient slabs then covers whole 128-byte segments (round 5; same values, same bits).
template <class Epi, class = void> struct epi_hi_off { static constexpr int value = 4; };
template <class Epi> struct epi_hi_off<Epi, std::void_t<decltype(Epi::HI_OFF)>> { static constexpr int value = Epi::HI_OFF; };
// ILV: how many of a phase's two LDS-DMA pieces are issued INSIDE the phase's MFMA segment instead of in front of its first barrier.
// Between two barriers one wave row runs its MFMA segment (8 MFMAs = 256 cycles + the fragment wait) while the other runs its
// load segment (fragment reads, 2 DMA issues at ~60-180 cycles each, the counted vmcnt wait); the barrier interval is the longer of
// the two, and with both DMA issues in the load segment that segment is the longer one (measured 435 cycles per interval against
// ~280 for the MFMA segment: 59 % matrix-pipe utilisation in the main loop).  Moving issues between the MFMAs balances the two.
// The counted wait shrinks by ILV (the pieces of this phase that are not issued yet do not count); every hazard distance of the
// header comment only grows (a unit is re-staged later, never earlier; it is still waited for >= 1 barrier before its first read).
// SHORT: 224 x 256 output tiles on the same workgroup (round 4).  12544 = 49 x 256 rows leave every R = 12544 launch with a last round
// of 256-row tiles that fills 6 % (fc1: 784 tiles on 256 CUs) to 77 % (proj / fc2: 196) of the chip; 12544 = 56 x 224 gives 896 / 224
// tiles of 7/8 the work each.  The tile keeps the 256-row LDS image and schedule; the lower wave row simply owns three 32-row blocks
// instead of four: its fourth block (tile rows 224..255, which belong to the next tile) is staged but never read, multiplied or
// stored -- 14 instead of 16 MFMAs per k-step on every SIMD (each SIMD hosts one wave of either row).  Same ascending K order per
// output element: bit-identical results.  Only for a K-major A operand (forward and data-gradient GEMMs).
constexpr int BM_SHORT = 224;
// MIXED (round 6): full-height tiles for a whole number of rounds of 256 workgroups, then HALF tiles of 128 rows for the rest of the rows.
// R = 12544 = 2^8 * 7^2 rows leave every multi-round GEMM of the ViT-L blocks with a last round that is half empty (fc1: 896 tiles of 224
// rows = 3.5 rounds, run as 4; qkv: 2.6 as 3; every uniform tiling was enumerated in round 3 -- docs/HISTORY.md 4.5).  A half tile keeps the
// workgroup and the 256-row LDS image, but runs its OWN loop (one top-level, workgroup-uniform branch; the full-tile loop is untouched):
// wave row r computes rows 64 r .. 64 r + 63 of the tile -- operand unit a_r, which it reads where wave row 0 reads its rows -- against both
// column halves: the MFMA segments of phases 0 / 1 of the full schedule, on 64 accumulator registers.  Phases 2 / 3 have no MFMAs; they
// re-stage the four units of the stage just read for the tile after next (all DMA of a K tile sits there, 8 pieces ahead of its first read:
// a deeper look-ahead than the full loop's).  A K tile so takes ~ 0.6 of a full tile's time for half its MFMAs.  (The DMA of the lower 128
// image rows is wasted on the next tile's rows: 25 % more L2 reads on 6 % of the tiles.)  fc1: 48 x 16 full tiles = 3 rounds exactly +
// 2 x 16 half tiles on idle CUs instead of a fourth round.  Same ascending K order per output element: bit-identical results.
// The first attempt predicated the MFMA segments of ONE loop per wave row (run-time branches inside the K loop): hipcc then drains vmcnt /
// lgkmcnt at every join -- the LDS-DMA and the asm fragment reads are opaque to it -- and the FULL tiles ran 22 - 28 % slower
// (profiles/r06_ab_mixed_tiles_first_attempt.log).  No branch may sit inside this kernel's K loop.
// Tiles [0, nfull * tiles_n) are full (BMR rows), the rest are half tiles starting at row nfull * BMR; only for a K-major A operand, no K split.
constexpr int BM_HALF = 128;
template <bool AMM, bool BMM, int ILV, class Epi, bool SHORT = false, bool MIXED = false>
__global__ __launch_bounds__(NT) void gemm256_kernel(const bf16* __restrict__ Ag, uint32_t lda, const bf16* __restrict__ Bg,
                                                      uint32_t ldb, Epi epi, int M, int N, int ktiles, int ktiles_per_split,
                                                      int tiles_n, int stagger, int order, int nfull, int patch) {
    static_assert(!(SHORT && AMM), "the 224-row tile is built for a K-major A operand");
    static_assert(!(MIXED && (AMM || SHORT)), "half tiles are built for a K-major A operand and 256-row full tiles");
    constexpr int BMR = SHORT ? BM_SHORT : BM;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    // de-phasing of the first round (diagnostic knob, pa_debug_set(0, cycles)): the 256 workgroups that start together are delayed by
    // 0 .. 15/16 of `stagger` so that their store phases do not hit HBM in one burst; later rounds inherit the phase of the CU they get
    if (stagger > 0 && blockIdx.x < 256 && blockIdx.y == 0) {
        const int64_t wait = ((int64_t)stagger * (((blockIdx.x >> 3) * 5) & 15)) >> 4;
        const uint64_t t0 = __builtin_amdgcn_s_memtime();
        while ((int64_t)(__builtin_amdgcn_s_memtime() - t0) < wait) __builtin_amdgcn_s_sleep(16);
    }
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wv >> 2, wc = wv & 3;

    // XCD-aware tile order: workgroup b runs on XCD b % 8; give each XCD a contiguous run of tiles (j fastest) so that the
    // tiles sharing an A row panel hit the same L2.
    // (MIXED: the full tiles and the half tiles are two regions of the grid, each ordered on its own)
    const bool htile = MIXED && (int)blockIdx.x >= nfull * tiles_n;
    const int nwg = MIXED ? (htile ? (int)gridDim.x - nfull * tiles_n : nfull * tiles_n) : (int)gridDim.x;
    const int bid = htile ? (int)blockIdx.x - nfull * tiles_n : (int)blockIdx.x;
    const int xq = nwg >> 3, xr = nwg & 7, xcd = bid & 7;
    int tile = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
    int split = blockIdx.y;
    // Split-K launches (the weight gradients; round 5): the workgroups are dispatched x-fastest over (tiles, splits) and land on XCD
    // (linear id) % 8.  With split = blockIdx.y every XCD used to hold an eighth of the tiles of EVERY split, so each of the 8 L2s pulled
    // all K slabs of its operand panels: the fc1 weight gradient read 431 MB where 129 are algorithmic (PMC, profiles/roofline_traffic.json).
    // Now the (split, tile) list -- split-major -- is cut into 8 contiguous runs, one per XCD: an XCD holds one split (or a few whole ones,
    // or a part of one) and inside it a compact patch of tiles, so a K slab of the operands goes through as few L2s as the sizes allow.
    // Needs the workgroup count to be a multiple of 8; order 2 (g_dbg[2]) keeps the old assignment for A/B.
    if (gridDim.y > 1 && order == 0 && ((nwg * gridDim.y) & 7) == 0) {
        const int lin = blockIdx.y * nwg + bid;
        const int w = (lin & 7) * ((nwg * (int)gridDim.y) >> 3) + (lin >> 3);
        split = w / nwg;
        tile = w - split * nwg;
    }
    // Inside the run the tiles are ordered in blocks of TR row panels x TC column panels (row groups of TR panels, column blocks of TC,
    // then row-major inside a block), so that the ~32 tiles an XCD has in flight form a TR x TC patch: per contraction step they pull
    // TR + TC operand panels through that XCD's L2 instead of 2 + tiles_n (fc1 forward, 16 column panels: FETCH_SIZE 251 -> ~165 MB
    // per launch; the weight matrix alone is twice the L2).  g_dbg[2] = 1 restores the plain row-major order (A/B).
    int tm, tn;
    if (order != 1) {
        const int TR = patch > 0 ? patch / 100 : 4, TC = patch > 0 ? patch % 100 : 8;      // patch (PA_G256_PATCH / pa_debug_set(11, TR * 100 + TC)): experiments
        const int tiles_m = MIXED ? (htile ? (M - nfull * BMR + BM_HALF - 1) / BM_HALF : nfull) : (M + BMR - 1) / BMR;
        const int per_group = TR * tiles_n;
        const int gm = tile / per_group, rem = tile - gm * per_group;
        const int rg = min(TR, tiles_m - gm * TR);                   // row panels in this (possibly last, shorter) group
        const int full = tiles_n / TC;                               // full column blocks
        int cb = rem / (rg * TC), r2 = rem - cb * (rg * TC), cw = TC;
        if (cb >= full) {                                            // the narrower last column block
            cb = full;
            r2 = rem - full * (rg * TC);
            cw = tiles_n - full * TC;
        }
        const int dm = r2 / cw;
        tm = gm * TR + dm;
        tn = cb * TC + (r2 - dm * cw);
    } else {
        tm = tile / tiles_n;
        tn = tile - tm * tiles_n;
    }
    const int i0 = htile ? nfull * BMR + tm * BM_HALF : tm * BMR, j0 = tn * BN;
    const bool blk3 = !(SHORT && wr == 1);          // does this wave own the fourth 32-row block of its 128 rows?  (wave-uniform)
    const int kt0 = split * ktiles_per_split;
    const int nt = min(ktiles - kt0, ktiles_per_split);

    // ---- DMA sources: 8 per lane (4 units x 2), as byte offsets from a wave-uniform, per-tile advancing base
    uint32_t oa[2][2], ob[2][2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            oa[sub][j] = src_off<AMM, true>(tid, j, sub, i0, M, lda);
            ob[sub][j] = src_off<BMM, false>(tid, j, sub, j0, N, ldb);
        }
    const size_t a_step = AMM ? (size_t)BK * lda * 2 : (size_t)BK * 2;
    const size_t b_step = BMM ? (size_t)BK * ldb * 2 : (size_t)BK * 2;
    const unsigned char* abase = reinterpret_cast<const unsigned char*>(Ag) + (size_t)kt0 * a_step;
    const unsigned char* bbase = reinterpret_cast<const unsigned char*>(Bg) + (size_t)kt0 * b_step;
    unsigned char* const dma_dst = smem + wv * 1024;

    // one 8 KB piece (j = 0, 1) of a 16 KB staging unit
    auto piece_a = [&](int sub, int stage, int kt, int j) {
        const unsigned char* s = sgpr_ptr(abase + (size_t)kt * a_step);
        dma16(s + oa[sub][j], dma_dst + unit_off(false, sub, stage) + j * 8192);
    };
    auto piece_b = [&](int sub, int stage, int kt, int j) {
        const unsigned char* s = sgpr_ptr(bbase + (size_t)kt * b_step);
        dma16(s + ob[sub][j], dma_dst + unit_off(true, sub, stage) + j * 8192);
    };
    auto stage_a = [&](int sub, int stage, int kt) { piece_a(sub, stage, kt, 0); piece_a(sub, stage, kt, 1); };
    auto stage_b = [&](int sub, int stage, int kt) { piece_b(sub, stage, kt, 0); piece_b(sub, stage, kt, 1); };

    // ---- fragment read bases (per lane)
    uint32_t ra[4], rb[4];
    {
        const int lr = lane & 31, g = lane >> 5;
        const int t = ((lr >> 1) & 7) ^ g;
        const int i = lane & 15, half = (lane >> 4) & 1;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            // (half tile: wave row r reads unit a_r -- 32 KB further on -- at the rows wave row 0 owns in it)
            if constexpr (!AMM) ra[ks] = (uint32_t)((htile ? wr * 32768 + lr * 128 : (wr * 64 + lr) * 128) + ((t << 4) ^ (ks << 5)));
            if constexpr (!BMM) rb[ks] = (uint32_t)(65536 + (wc * 32 + lr) * 128 + ((t << 4) ^ (ks << 5)));
        }
        if constexpr (AMM) {
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                const int c = wr * 8 + mb * 4 + half * 2 + ((i & 3) >> 1);
                ra[mb] = (uint32_t)((g * 8 + (i >> 2)) * 256 + ((c ^ (4 * (i >> 2))) << 4) + (i & 1) * 8);
            }
            ra[2] = ra[3] = 0;
        }
        if constexpr (BMM) {
            const int c = wc * 4 + half * 2 + ((i & 3) >> 1);
            rb[0] = (uint32_t)(65536 + (g * 8 + (i >> 2)) * 256 + ((c ^ (4 * (i >> 2))) << 4) + (i & 1) * 8);
            rb[1] = rb[2] = rb[3] = 0;
        }
    }

    f32x16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    Frag4<AMM> fa0, fa1;
    Frag4<BMM> fb0, fb1;

    // the MFMA takes the B fragment first so that a lane ends up with ONE output row and 4-column runs (wide stores)
#define G256_MMA(KS)                                                                                                     \
    acc[mrow * 2 + 0][ncol] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr.template k<KS>(), fa0.template k<KS>(), acc[mrow * 2 + 0][ncol], 0, 0, 0); \
    if (!SHORT || mrow == 0 || blk3)                                                                                     \
        acc[mrow * 2 + 1][ncol] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr.template k<KS>(), fa1.template k<KS>(), acc[mrow * 2 + 1][ncol], 0, 0, 0);
    // d0 / d1: this phase's DMA pieces; issued between the MFMAs (order pinned) as far as ILV says, otherwise by the caller
    auto mma_quad = [&](int mrow, int ncol, const Frag4<BMM>& bfr, auto&& d0, auto&& d1) {
        __builtin_amdgcn_s_setprio(1);
        G256_MMA(0)
        if constexpr (ILV >= 2) { __builtin_amdgcn_sched_barrier(0); d0(); __builtin_amdgcn_sched_barrier(0); }
        G256_MMA(1) G256_MMA(2)
        if constexpr (ILV >= 1) { __builtin_amdgcn_sched_barrier(0); d1(); __builtin_amdgcn_sched_barrier(0); }
        G256_MMA(3)
        __builtin_amdgcn_s_setprio(0);
    };
#undef G256_MMA
    // the part of a phase's DMA that stays in front of its first barrier, and the counted wait that goes with it
    auto pre = [&](auto&& d0, auto&& d1) {
        if constexpr (ILV < 2) d0();
        if constexpr (ILV < 1) d1();
        wait_vm<8 - ILV>();
    };
    auto bar = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto lwait = [&]() {
        wait_lgkm0();
        __builtin_amdgcn_sched_barrier(0);
    };

    // DMA targets past the last tile are clamped to it: the unit lands in a slot nobody reads again, and the wait counts
    // stay uniform (no tail variants of the loop body, which would make hipcc spill around the asm reads).
    auto tile_body = [&](auto stage_c, int T) {
        constexpr int S = decltype(stage_c)::value;
        const int t1 = min(T + 1, nt - 1), t2 = min(T + 2, nt - 1);
        // ---- phase 0: a0, b0
        fa0.template read<unit_off(false, 0, S)>(ra, 0);
        fa1.template read<unit_off(false, 0, S) + (AMM ? 0 : 4096)>(ra, 1);
        fb0.template read<unit_off(true, 0, S) - 65536>(rb, 0);
        {
            auto d0 = [&] { piece_b(1, S ^ 1, t1, 0); };
            auto d1 = [&] { piece_b(1, S ^ 1, t1, 1); };
            pre(d0, d1);
            bar(); lwait();
            mma_quad(0, 0, fb0, d0, d1);
            bar();
        }
        // ---- phase 1: b1
        fb1.template read<unit_off(true, 1, S) - 65536>(rb, 0);
        {
            auto d0 = [&] { piece_a(1, S ^ 1, t1, 0); };
            auto d1 = [&] { piece_a(1, S ^ 1, t1, 1); };
            pre(d0, d1);
            bar(); lwait();
            mma_quad(0, 1, fb1, d0, d1);
            bar();
        }
        // ---- phase 2: a1
        fa0.template read<unit_off(false, 1, S)>(ra, 0);
        if (!SHORT || blk3) fa1.template read<unit_off(false, 1, S) + (AMM ? 0 : 4096)>(ra, 1);
        {
            auto d0 = [&] { piece_a(0, S, t2, 0); };
            auto d1 = [&] { piece_a(0, S, t2, 1); };
            pre(d0, d1);
            bar(); lwait();
            mma_quad(1, 1, fb1, d0, d1);
            bar();
        }
        // ---- phase 3: nothing new to read
        {
            auto d0 = [&] { piece_b(0, S, t2, 0); };
            auto d1 = [&] { piece_b(0, S, t2, 1); };
            pre(d0, d1);
            bar();
            mma_quad(1, 0, fb0, d0, d1);
            bar();
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    // half tile (MIXED): a K tile is TWO phases -- the MFMA segments of phases 0 / 1 of the schedule above, on unit a_wr -- with ONE barrier
    // each, all eight waves in step (no half-phase offset between the wave rows: with two phases per tile the offset would need the DMA
    // waited two barriers ahead of its first read, i.e. inside the interval it is issued in).  The DMA of the NEXT tile goes into the other
    // stage -- whose last reads completed in front of the previous barrier -- between this tile's MFMAs: four pieces in phase 0 (the
    // upper 8 KB halves of a0 / a1 -- the lower halves hold image rows 128..255, which a half tile never reads -- and b0), two in phase 1
    // (b1).  Counted waits at the END of a phase, in front of its barrier: phase 0 leaves its own four pieces in flight (b1 of this tile has
    // landed: read right behind the barrier), phase 1 its own two (a0, a1, b0 of the next tile have landed).
    // (The first version kept the four-phase frame and re-staged in two MFMA-free phases: eight bare DMA issues cost as much as the MFMA
    // segments they no longer hid behind -- a half round took 0.85 of a full round: profiles/r06_ab_mixed_tiles_half_loop.log.)
    auto half_body = [&](auto stage_c, int T) {
        constexpr int S = decltype(stage_c)::value;
        const int t1 = min(T + 1, nt - 1);
#define G256_HMMA(KS, NC) \
        acc[0][NC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16((NC ? fb1 : fb0).template k<KS>(), fa0.template k<KS>(), acc[0][NC], 0, 0, 0); \
        acc[1][NC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16((NC ? fb1 : fb0).template k<KS>(), fa1.template k<KS>(), acc[1][NC], 0, 0, 0);
#define G256_HDMA(X) __builtin_amdgcn_sched_barrier(0); X; __builtin_amdgcn_sched_barrier(0);
        // ---- phase 0: a_wr, b0 of stage S; stage the next tile's a0, a1 (upper halves), b0 into stage S ^ 1
        fa0.template read<unit_off(false, 0, S)>(ra, 0);
        fa1.template read<unit_off(false, 0, S) + 4096>(ra, 1);
        fb0.template read<unit_off(true, 0, S) - 65536>(rb, 0);
        lwait();
        __builtin_amdgcn_s_setprio(1);
        G256_HMMA(0, 0) G256_HDMA(piece_a(0, S ^ 1, t1, 0))
        G256_HMMA(1, 0) G256_HDMA(piece_a(1, S ^ 1, t1, 0))
        G256_HMMA(2, 0) G256_HDMA(piece_b(0, S ^ 1, t1, 0))
        G256_HMMA(3, 0) G256_HDMA(piece_b(0, S ^ 1, t1, 1))
        __builtin_amdgcn_s_setprio(0);
        wait_vm<4>();
        bar();
        // ---- phase 1: b1 of stage S; stage the next tile's b1
        fb1.template read<unit_off(true, 1, S) - 65536>(rb, 0);
        lwait();
        __builtin_amdgcn_s_setprio(1);
        G256_HMMA(0, 1) G256_HDMA(piece_b(1, S ^ 1, t1, 0))
        G256_HMMA(1, 1) G256_HMMA(2, 1) G256_HDMA(piece_b(1, S ^ 1, t1, 1))
        G256_HMMA(3, 1)
        __builtin_amdgcn_s_setprio(0);
        wait_vm<2>();
        bar();
#undef G256_HMMA
#undef G256_HDMA
    };

    bool ran_half = false;
    if constexpr (MIXED) {
        if (htile) {                       // workgroup-uniform, outside every loop
            ran_half = true;
            piece_a(0, 0, 0, 0); piece_a(1, 0, 0, 0); stage_b(0, 0, 0); stage_b(1, 0, 0);      // tile 0 into stage 0
            wait_vm<0>();
            bar();
            for (int T = 0; T < nt; T += 2) {
                half_body(I0{}, T);
                half_body(I1{}, T + 1);
            }
        }
    }
    if (!ran_half) {
        // ---- prologue: tile 0 entirely, plus a0/b0 of tile 1   (nt is even and >= 2: see launch())
        stage_a(0, 0, 0);
        stage_b(0, 0, 0);
        stage_b(1, 0, 0);
        stage_a(1, 0, 0);
        stage_a(0, 1, 1);
        stage_b(0, 1, 1);
        wait_vm<8>();
        bar();
        if (wr == 1) bar();   // lower wave row runs half a phase behind

        for (int T = 0; T < nt; T += 2) {
            tile_body(I0{}, T);
            tile_body(I1{}, T + 1);
        }
    }
    if (wr == 0 && !ran_half) bar();       // (the full loop's half-phase offset between the wave rows; the half-tile loop runs in step)
    wait_vm<0>();          // the clamped tail DMAs must have landed before this workgroup's LDS is handed on

    // ---- epilogue.  A lane owns one output row of each 32x32 block; storing from there would touch 32 cache lines per
    // store instruction (measured: the store tail cost ~1/3 of a K=1024 tile).  Instead every wave passes its tile, one
    // 32 x 64 block at a time, through a private LDS scratch ([32][68] floats, padded: conflict-free ds_write_b128) and
    // re-reads it 8 columns per lane, 8 lanes per row, so the fused epilogue stores whole 128/256-byte row segments.
    // Global LOADS of the epilogue (bias, residual, GELU pre-activation) are issued ahead of the stores they feed: a load
    // placed after a store can only be waited for with vmcnt(0), i.e. together with every store still in flight (stores and
    // loads share the counter), which made each of the 16 row groups pay a full store round trip (rocprof: 11-13 us of a
    // 41 us fc1 tile).  col() is loaded once per tile, row() for two 32-row blocks at a time before that half's stores.
    bar();                 // every wave's DMAs have landed and nobody reads the staging units any more
    {
        float* stg = reinterpret_cast<float*>(smem + wv * 8704);
        constexpr int HO = epi_hi_off<Epi>::value;
        static_assert(HO == 4 || (HO == 32 && !epi_colsum<Epi>::value), "epilogue column layouts: 8 consecutive, or 4 + 4 at a distance of 32");
        const int lr = lane & 31, g = lane >> 5, rrow = lane >> 3, c0 = (lane & 7) * (HO == 4 ? 8 : 4);
        const int jcol = j0 + wc * 64 + c0;
        const typename Epi::Col col = epi.col(jcol);
        const int rbase = i0 + wr * (htile ? 64 : 128);         // first tile row of this wave
        float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            if (MIXED && htile && hf != 0) continue;           // half tile: a wave holds two 32-row blocks (accumulator blocks 0, 1): rows 64 wr .. 64 wr + 63
            typename Epi::Row rows[2][4];
#pragma unroll
            for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    const int mb = hf * 2 + m2;
                    if (SHORT && mb == 3 && !blk3) continue;
                    rows[m2][st] = epi.row(rbase + (mb >> 1) * 64 + (mb & 1) * 32 + st * 8 + rrow, jcol);
                }
#pragma unroll
            for (int m2 = 0; m2 < 2; ++m2) {
                const int mb = hf * 2 + m2;
                if (SHORT && mb == 3 && !blk3) continue;
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x16& c = acc[mb][nb];
                        *reinterpret_cast<float4*>(stg + lr * 68 + nb * 32 + q * 8 + g * 4) =
                            make_float4(c[q * 4 + 0], c[q * 4 + 1], c[q * 4 + 2], c[q * 4 + 3]);
                    }
                const int ib = rbase + (mb >> 1) * 64 + (mb & 1) * 32;
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    const int r = st * 8 + rrow;
                    const float4 lo = *reinterpret_cast<const float4*>(stg + r * 68 + c0);
                    const float4 hi = *reinterpret_cast<const float4*>(stg + r * 68 + c0 + HO);
                    if constexpr (epi_colsum<Epi>::value) epi.store_cs(ib + r, jcol, lo, hi, col, rows[m2][st], split, cs);
                    else epi.store(ib + r, jcol, lo, hi, col, rows[m2][st], split);
                }
            }
        }
        if constexpr (epi_colsum<Epi>::value) {
            // the 8 lanes l, l + 8, ..., l + 56 hold the same 8 columns (other rows): rotate-by-8 inside a row of 16 lanes, then across
            // rows with the permlane swaps; lanes 0..7 end up with the wave's sums and write the partial row of (row tile, wave row)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = cs[e];
                v += lane_dpp<0x128>(v);
                v += lane_xor16(v);
                v += lane_xor32(v);
                cs[e] = v;
            }
            if (lane < 8) epi.colsum_out((htile ? nfull + tm : tm) * 2 + wr, jcol, cs);      // one partial row per (row tile, wave row), full tiles first
        }
    }
}

static inline int per_split(int ktiles, int nsplit) {
    int per = (ktiles + nsplit - 1) / nsplit;
    return per + (per & 1);
}
template <bool AMM, bool BMM, int ILV, bool SHORT, class Epi, bool MIXED = false>
static int launch_ilv(const bf16* A, size_t lda, const bf16* B, size_t ldb, Epi epi, int M, int N, int K, int nsplit, hipStream_t st, int nfull = 0);
#ifndef G256_ILV_DEFAULT
#define G256_ILV_DEFAULT 2     // round 3 (tools/gemm_ilv_ab.py, MI355X): 2 is 4-15 % faster than 0 on the forward GEMMs, 4-6 % on the data gradients,
#endif                         // 2 % on the weight gradients, bit-identical results; 1 = 0.  In the training step the gain shrinks to ~1 % (DVFS)
// Which row-tile height for an un-split launch with a K-major A operand: rounds of 256 workgroups x the tile's relative cost.
// g_dbg[4] (pa_debug_set(4, v)): 0 = this rule, 1 = always 256 rows, 2 = 224 rows wherever the kernel can.  PA_G256_SHORT_COST: the
// 224-row tile's cost relative to the 256-row tile in percent (default 92: 14 of 16 MFMAs, the full load segment and barriers).
static inline bool use_short(int M, int N, int nsplit, bool amm) {
    if (amm || nsplit != 1 || M < BM) return false;
    if (g_dbg[4] == 1) return false;
    if (g_dbg[4] == 2) return true;
    static const int cost = [] { const char* v = getenv("PA_G256_SHORT_COST"); return v ? atoi(v) : 92; }();
    const int tiles_n = (N + BN - 1) / BN;
    const long t256 = (long)((M + BM - 1) / BM) * tiles_n, t224 = (long)((M + BM_SHORT - 1) / BM_SHORT) * tiles_n;
    const long r256 = (t256 + 255) / 256 * 100, r224 = (t224 + 255) / 256 * cost;
    return r224 < r256;
}
// Tile plan of an un-split launch with a K-major A operand (round 6): uniform 256-row tiles, uniform 224-row tiles, or MIXED = `nfull` rows of
// 256-row tiles filling whole rounds of 256 workgroups + half tiles (128 rows) for the remaining rows.  Costs in percent of a 256-row round,
// measured (tools/gemm_mixed_probe.py, profiles/r06_ab_mixed_tiles_half_loop.log): a 224-row round 97 -- 14 of 16 MFMAs, the whole load segment
// and every barrier (PA_G256_SHORT_COST; the choice between the two uniform tilings keeps round 4's 92) --, a half-tile round 75
// (PA_G256_HALF_COST): half the MFMAs on 3/4 of the LDS-DMA, and the DMA issues are what the loop is short of.  pa_debug_set(12, v): 0 = this
// rule, 1 = no mixed plans (the round-5 behaviour), 2 = mixed with pa_debug_set(14, nfull) full rows (diagnostics).
struct TilePlan { bool is_short, mixed; int nfull; };
static inline TilePlan tile_plan(int M, int N, int nsplit, bool amm) {
    TilePlan p{use_short(M, N, nsplit, amm), false, 0};
    if (amm || nsplit != 1 || M < 2 * BM || g_misc_knob[1] == 1 || g_dbg[4] != 0) return p;
    if (g_misc_knob[1] == 2 && g_misc_knob[3] > 0 && (long)g_misc_knob[3] * BM < M) return TilePlan{false, true, g_misc_knob[3]};      // diagnostics (tools/gemm_mixed_probe.py): pa_debug_set(12, 2) + pa_debug_set(14, nfull)
    static const int cs = [] { const char* v = getenv("PA_G256_SHORT_COST"); return v ? atoi(v) : 97; }();
    static const int ch = [] { const char* v = getenv("PA_G256_HALF_COST"); return v ? atoi(v) : 75; }();
    const int tn = (N + BN - 1) / BN;
    auto rounds = [](long t) { return (t + 255) / 256; };
    long best = p.is_short ? rounds((long)((M + BM_SHORT - 1) / BM_SHORT) * tn) * cs : rounds((long)((M + BM - 1) / BM) * tn) * 100;
    // (only 256-row full tiles; as many whole rounds of them as the rows allow, i.e. as few half tiles as possible: `<=` below)
    for (int r = 1; r <= 64; ++r) {
        const int nf = (int)((long)r * 256 / tn);
        if (nf < 1) continue;
        if ((long)nf * BM >= M) break;                                  // covered by full tiles alone: the uniform plans above
        const long nh = (M - (long)nf * BM + BM_HALF - 1) / BM_HALF;
        const long c = (long)r * 100 + rounds(nh * tn) * ch;
        if (c <= best - 5) { best = c + 5; p = TilePlan{false, true, nf}; }      // at least 5 % of a round better than what it replaces; later r (more full rows) wins ties
    }
    return p;
}
template <bool AMM, bool BMM, class Epi>
static int launch(const bf16* A, size_t lda, const bf16* B, size_t ldb, Epi epi, int M, int N, int K, int nsplit,
                  hipStream_t st) {
    if constexpr (!AMM) {
        const TilePlan tp = tile_plan(M, N, nsplit, AMM);
        if (tp.mixed) return launch_ilv<AMM, BMM, G256_ILV_DEFAULT, false, Epi, true>(A, lda, B, ldb, epi, M, N, K, nsplit, st, tp.nfull);
        if (tp.is_short) return launch_ilv<AMM, BMM, G256_ILV_DEFAULT, true>(A, lda, B, ldb, epi, M, N, K, nsplit, st);
    }
#ifdef G256_ILV_AB       // experiment build: all three schedules in one library, pa_debug_set(5, 1 + ILV) picks one at run time
    static const int env_ilv = [] { const char* v = getenv("PA_G256_ILV"); return v ? atoi(v) : G256_ILV_DEFAULT; }();
    const int ilv = g_dbg[5] > 0 ? g_dbg[5] - 1 : env_ilv;
    if (ilv == 2) return launch_ilv<AMM, BMM, 2, false>(A, lda, B, ldb, epi, M, N, K, nsplit, st);
    if (ilv == 1) return launch_ilv<AMM, BMM, 1, false>(A, lda, B, ldb, epi, M, N, K, nsplit, st);
    return launch_ilv<AMM, BMM, 0, false>(A, lda, B, ldb, epi, M, N, K, nsplit, st);
#else
    return launch_ilv<AMM, BMM, G256_ILV_DEFAULT, false>(A, lda, B, ldb, epi, M, N, K, nsplit, st);
#endif
}
template <bool AMM, bool BMM, int ILV, bool SHORT, class Epi, bool MIXED>
static int launch_ilv(const bf16* A, size_t lda, const bf16* B, size_t ldb, Epi epi, int M, int N, int K, int nsplit, hipStream_t st, int nfull) {
    auto kern = gemm256_kernel<AMM, BMM, ILV, Epi, SHORT, MIXED>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    constexpr int BMR = SHORT ? BM_SHORT : BM;
    static const int env_patch = [] { const char* v = getenv("PA_G256_PATCH"); return v ? atoi(v) : 0; }();
    const int tiles_n = (N + BN - 1) / BN;
    const int tiles_m = MIXED ? nfull + (M - nfull * BMR + BM_HALF - 1) / BM_HALF : (M + BMR - 1) / BMR;
    const int ktiles = K / BK;
    const int per = per_split(ktiles, nsplit);
    const int splits = (ktiles + per - 1) / per;      // every split gets an even number (>= 2) of tiles
    if (g_dbg[1]) epi.M = 0;
    PA_LAUNCH(kern, dim3(tiles_m * tiles_n, splits), dim3(NT), LDS_BYTES, st, A, (uint32_t)lda, B, (uint32_t)ldb, epi, M, N,
              ktiles, per, tiles_n, g_dbg[0], g_dbg[2], nfull, g_misc_knob[0] > 0 ? g_misc_knob[0] : env_patch);
    return (int)hipGetLastError();
}
// row tiles of the launch launch<AMM = false>(...) makes for this shape (partial rows of a column-sum epilogue = 2 x this)
static inline int row_tiles_used(int M, int N, int nsplit) {
    const TilePlan tp = tile_plan(M, N, nsplit, false);
    const int bm = tp.is_short ? BM_SHORT : BM;
    if (tp.mixed) return tp.nfull + (M - tp.nfull * bm + BM_HALF - 1) / BM_HALF;
    return (M + bm - 1) / bm;
}
// shapes the kernel accepts; everything else stays on the generic engine (gemm_engine.h)
static inline bool ok(int M, int N, int K, bool amm, bool bmm, size_t lda, size_t ldb) {
    if (K % (2 * BK) || M < 8 || N < 8 || N % 8) return false;
    if (amm && (M % 8)) return false;
    if (bmm && (N % 8)) return false;
    if ((lda % 8) || (ldb % 8)) return false;
    // 32-bit byte offsets inside one operand
    const size_t abytes = (amm ? (size_t)BK * lda + M : (size_t)M * lda) * 2, bbytes = (bmm ? (size_t)BK * ldb + N : (size_t)N * ldb) * 2;
    return abytes < (1ull << 32) && bbytes < (1ull << 32);
}
static inline int splits_used(int K, int nsplit) {
    const int ktiles = K / BK, per = per_split(ktiles, nsplit);
    return (ktiles + per - 1) / per;
}

}   // namespace g256
