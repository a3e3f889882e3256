// bf16 fused attention backward, fourth generation: 64-row waves.  Same math, table tiles, one-hot contraction and C ABI as attn3.hip
// (Painter/models_painter.py:76-86, util/vitdet_utils.py:63-125); only the work split differs.
//
// Why.  Generation 3 runs two 32-row waves per SIMD (two workgroups per CU).  Its counters (profiles/r03_attn_sq_counters.json): matrix
// pipe 0.34-0.38 busy, VALU 0.37, waves parked 31 % and issue-stalled 34-41 % of their cycles -- nothing saturated; a wave's tile is the
// serial chain {LDS fragment wait -> 10 MFMAs -> softmax VALU -> 6 MFMAs -> barrier} and the two waves of a SIMD, which belong to
// different workgroups, overlap it only by chance.  Here ONE wave per SIMD owns TWO 32-row blocks (64 query rows in dQ, 64 keys in dKV):
//   * every K / V / one-hot (dQ) or Q / dO / table (dKV) fragment read from LDS feeds two blocks: half the LDS reads per MFMA;
//   * block A's MFMAs and block B's softmax arithmetic are independent instructions of one stream, so the scheduler interleaves them
//     (matrix pipe and VALU run side by side by construction instead of by the luck of two waves' relative phase);
//   * the whole 512-register file belongs to the wave: no spills, -Delta rides in the accumulator init for both blocks.
// A workgroup = 4 waves = 8 blocks = 256 rows of the stationary axis; LDS is sized for one workgroup per CU.  1568 tokens = 49 tiles =
// 6 workgroups x 8 + 1: the kernels here take the whole 8-tile groups (6 x 128 heads = 768 workgroups = exactly three rounds of 256 CUs
// at the ViT-L B = 8 shape), the 49th tile of every head runs on the generation-3 kernels (attn3.hip, `tile0` launches).
// The rel-pos table gradient is always contracted in the dQ kernel (attn3.hip, FUSE): one fp32 [NRP][64] partial per workgroup.
//
// STATUS (round 4, MI355X): dQ is built, parity-green (tests/test_kernels_gpu.py::test_attn4_64_row_backward_vs_generation_3_and_fp64)
// and NOT the default: at the ViT-L shape the 768-workgroup launch takes 235 us + 54 us for the 49th tiles on generation 3, against
// 206 us for generation 3 alone.  s_memtime stamps of the hand-ordered loop below (tools/attn4_lab.py, experiment build): ~3100 cycles
// per pair of 32 x 32 tiles = top (staging stores, loads, window reads) 370 | G1 390 | G2 550 | G3 540 | barrier 240 | G4 + LDS
// returns 640 | window write-back 370, against 2260 for two generation-3 waves: a single in-order wave overlaps only what the program
// order interleaves, every MFMA result the VALU touches costs a v_accvgpr_read here (the compiler selects the AGPR form of every MFMA
// once a kernel may use more than 256 registers: 64 extra VALU instructions per tile pair), block B's softmax (88 VALU) has only six
// MFMAs to hide under, and prologue / epilogue (~9 us per workgroup) are exposed with one workgroup per CU.  What it would take:
// top / tail folded into the MFMA gaps and a skew of one tile between the blocks (block B's softmax under block A's next S / dP chain).
// The 64-key dKV kernel was not written.
#include "attn3_common.h"
#include "../../include/painter_hip.h"
#include "attn3.h"
#include <cstdlib>

namespace a4 {
using namespace a3;
__device__ unsigned long long g_trace4[64 * 8];
DEVI bf16x8 kfr_init() { return __builtin_bit_cast(bf16x8, zero4()); }

constexpr int QB = 2;                 // 32-row blocks per wave
constexpr int WGB = NW * QB;          // blocks per workgroup
constexpr int NSMAX = 12;             // r-space steps held in registers (NRP <= 192)

// ---- LDS map of the dQ kernel (bytes).  Nothing aliases during the key loop and the r-space steps; the fused rel-pos contraction at
// the very end reuses the allocation from offset 0 (WGB x (nimg + 1) images of 4 KB).
struct DqLds {
    int kv, eimg, tht, rimg, twg, stg, trash, total;
    __host__ __device__ DqLds(int Hp, int NRP) {
        kv = 0;
        eimg = kv + 2 * STAGE_QK;
        tht = eimg + PH * EIMG;
        rimg = tht + WGB * Hp * 64;
        twg = rimg + ATT_HD * (NRP * 2 + 16);
        stg = twg + WGB * 32 * WP * 4;
        trash = stg + WGB * IMG;                    // 16 bytes per thread that nobody reads (branch-free stores: window write-back, prologue)
        total = trash + NT * 16;
        const int fused = WGB * ((NRP + 63) / 64 + 1) * IMG;
        if (total < fused) total = fused;
    }
};

// =============================================================================================== backward: dQ, bias gradients, d rel_pos partials
// grid: ngrp 8-tile query groups per head x (batch * heads); every block of every workgroup is live (whole groups only)
// LAB (diagnostics, compile-time ablation of the key loop, PA_ATTN4_DQ_LAB; results WRONG unless 0): 1 no exp / fma (dS = dP), 2 no global
// loads in the loop, 4 no barrier in the loop, 8 no LDS fragment reads in the loop, 16 no staging stores in the loop, 32 no MFMAs in the loop
template <int LAB>
__global__ __launch_bounds__(NT, 1) void bwd_dq64_kernel(const bf16* __restrict__ qkv, size_t ldq, const bf16* __restrict__ rcatT,
                                                         const bf16* __restrict__ dout, size_t lddo, const float* __restrict__ lse,
                                                         const unsigned char* __restrict__ tables, bf16* __restrict__ dqkv,
                                                         float* __restrict__ part, int L, int H, int Hp, int NRP, float scale, int ngrp,
                                                         int xcd_map, int abl) {
    // abl (diagnostics, PA_ATTN4_DQ_ABL; results WRONG when set): 16 no key loop, 1 no r-space steps, 2 no rel-pos contraction
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 5, ql = lane & 31;
    const DqLds lds(Hp, NRP);
    const int v = (xcd_map & 1) ? xcd_run(blockIdx.x, gridDim.x) : blockIdx.x;
    const int bh = v / ngrp, grp = v - bh * ngrp;
    const int b = bh / H, h = bh % H, D = H * ATT_HD;
    const bf16* base = qkv + (size_t)b * L * ldq + h * ATT_HD;
    const bf16* kbase = base + D;
    const bf16* vbase = base + 2 * D;
    const int ntile = L / 32;
    const float sl = scale * LOG2E_F;
    unsigned char* eimg = smem + lds.eimg;
    unsigned char* rimg = smem + lds.rimg;
    const int rpitch = NRP * 2 + 16;
    LaneAddr la;
    la.init(lane);
    EAddr ea;
    ea.init(lane);
    Stager ks, vs;
    ks.load(kbase, ldq, tid);
    vs.load(vbase, ldq, tid);

    unsigned char* const trash = smem + lds.trash + tid * 16;
    // ---- per-block state.  Every global load of the prologue is issued before the first LDS store (one memory round trip instead of
    // a dozen dependent ones: with one workgroup per CU nothing else covers them)
    int qt[QB], q[QB];
    bf16x8 qf[QB][4], dof[QB][4];
    uint4 T0[QB], T1[QB];
    float nlse2[QB];
    f32x16 ndl[QB];
    unsigned char* thT[QB];
    constexpr int TCH = 4;                               // 16-byte chunks of a kh table per lane (Hp * 4 <= 256 chunks: Hp <= 64)
    uint4 tch[QB][TCH];
    float ndlt[QB];
#pragma unroll
    for (int X = 0; X < QB; ++X) {
        qt[X] = grp * WGB + wave * QB + X;
        q[X] = qt[X] * 32 + ql;
        thT[X] = smem + lds.tht + (wave * QB + X) * Hp * 64;
        const unsigned char* tt = tables + ((size_t)bh * ntile + qt[X]) * ttile_bytes(Hp);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            qf[X][s] = gfrag(base + (size_t)q[X] * ldq, s, g);
            dof[X][s] = gfrag(dout + (size_t)(b * L + q[X]) * lddo + h * ATT_HD, s, g);
        }
        T0[X] = *reinterpret_cast<const uint4*>(tt + ql * 64 + 16 * g);
        T1[X] = *reinterpret_cast<const uint4*>(tt + ql * 64 + 32 + 16 * g);
        nlse2[X] = -lse[(size_t)bh * L + q[X]] * LOG2E_F;
        ndlt[X] = *reinterpret_cast<const float*>(tt + 2048 + (Hp + 2) * 64 + ql * 4);
#pragma unroll
        for (int i = 0; i < TCH; ++i) {
            const int c = min(lane + 64 * i, Hp * 4 - 1);      // clamped: a static number of loads, the extra ones are not stored
            tch[X][i] = *reinterpret_cast<const uint4*>(tt + 2048 + c * 16);
        }
    }
    constexpr int RCH = 6;                               // 16-byte chunks of Rcat^T per thread: thread = (row d = tid / 4, chunks (tid & 3) + 4 i), NRP <= 192
    uint4 rch[RCH];
    const int per_row = NRP / 8, rrow = tid >> 2;
#pragma unroll
    for (int i = 0; i < RCH; ++i) rch[i] = *reinterpret_cast<const uint4*>(rcatT + ((size_t)rrow * per_row + min((tid & 3) + 4 * i, per_row - 1)) * 8);
    __builtin_amdgcn_sched_barrier(0);                   // all of the above is in flight before anything below waits
    build_eimg(eimg, tid);
#pragma unroll
    for (int X = 0; X < QB; ++X) {
#pragma unroll
        for (int r = 0; r < 16; ++r) ndl[X][r] = ndlt[X];
#pragma unroll
        for (int i = 0; i < TCH; ++i)         // (out-of-range chunks go to the thread's trash slot: a conditional store would drag its load into the branch)
            *reinterpret_cast<uint4*>(lane + 64 * i < Hp * 4 ? thT[X] + (lane + 64 * i) * 16 : trash) = tch[X][i];
    }
#pragma unroll
    for (int i = 0; i < RCH; ++i) {                      // Rcat^T [64 d][NRP] -> LDS, row pitch NRP * 2 + 16 bytes (attn3.hip)
        const int col = (tid & 3) + 4 * i;
        *reinterpret_cast<uint4*>(col < per_row ? rimg + rrow * rpitch + col * 16 : trash) = rch[i];
    }
    // ---- key loop: an explicit software pipeline (one in-order wave per SIMD: nothing overlaps unless the program order says so).
    //   top      registers (tile j + 1, loaded during iteration j - 1) -> LDS stage (j + 1) & 1; the same registers then fetch tile j + 2
    //   G1       S / dP chains of block A (10 MFMAs)      | the transposed K / one-hot fragments of tile j in the gaps
    //   G2       S / dP chains of block B (10 MFMAs)      | block A's softmax arithmetic, two elements per gap
    //   G3       dQ / bias-gradient MFMAs of block A (6)  | block B's softmax arithmetic (what does not fit is exposed)
    //   barrier  every wave is done with tile j's LDS image, every wave's stores of tile j + 1 have landed
    //   G4       dQ / bias-gradient MFMAs of block B (6)  | the ten row fragments of tile j + 1 -> registers for the next iteration
    // sched_barrier(0) fences pin this order (the scheduler otherwise groups the MFMAs and the VALU work apart again).
#define A4_FENCE() __builtin_amdgcn_sched_barrier(0)
#define A4_STAMP(pt) do { if constexpr (LAB & 64) { A4_FENCE(); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (blockIdx.x == 0 && tid == 0 && j < 64) g_trace4[j * 8 + (pt)] = t_; A4_FENCE(); } } while (0)
    ks.store(smem + lds.kv, tid);
    vs.store(smem + lds.kv + IMG, tid);
    ks.load(kbase + (size_t)min(1, ntile - 1) * 32 * ldq, ldq, tid);
    vs.load(vbase + (size_t)min(1, ntile - 1) * 32 * ldq, ldq, tid);
    __syncthreads();

    // The four window slots of eacc (D rows 22, 23, 30, 31: registers of half-wave 1) are never cleared: a slot's accumulator is a
    // running sum over the key rows that used it, and the gradient of a completed row is the difference to the sum at the slot's
    // previous completion (kept per lane in wprev).  Clearing the register instead costs a round trip of the WHOLE accumulator
    // between the accumulator file and the VGPRs under an exec mask -- 64 moves per tile in this 512-register kernel.
    f32x16 dq[QB][2], eacc[QB];
    float wprev[QB][4];
#pragma unroll
    for (int X = 0; X < QB; ++X) {
        dq[X][0] = zero16();
        dq[X][1] = zero16();
        eacc[X] = zero16();
#pragma unroll
        for (int sl4 = 0; sl4 < 4; ++sl4) wprev[X][sl4] = 0.f;
    }
    bf16x8 ktr[2][2] = {{kfr_init(), kfr_init()}, {kfr_init(), kfr_init()}}, etr[2] = {kfr_init(), kfr_init()};
    // row fragments of the current tile (read one iteration ahead)
    bf16x8 ef0 = efrag(eimg, ea, 0), ef1 = efrag(eimg, ea, 1), kfr[4], vfr[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) { vfr[s] = rowfrag(smem + lds.kv + IMG, la, s); kfr[s] = rowfrag(smem + lds.kv, la, s); }

#define A4_TR(img, la, db, ks_) ((LAB & 8) ? ktr[db][ks_] : trfrag(img, la, db, ks_))
#define A4_ETR(img, ea, ks_) ((LAB & 8) ? etr[ks_] : etrfrag(img, ea, ks_))
#define A4_EF(img, ea, ks_) ((LAB & 8) ? ef0 : efrag(img, ea, ks_))
#define A4_ROW(img, la, ks_) ((LAB & 8) ? kfr[ks_] : rowfrag(img, la, ks_))
#define A4_MFMA(a_, b_, c_) ((LAB & 32) ? (c_) : mfma(a_, b_, c_))
    auto body = [&](auto pc, int a) {
        constexpr int P = decltype(pc)::value;
        const int j = a * PH + P;
        const unsigned char* kimg = smem + lds.kv + (j & 1) * STAGE_QK;
        unsigned char* nimg_kv = smem + lds.kv + ((j + 1) & 1) * STAGE_QK;
        A4_STAMP(0);
        // ---- top
        if constexpr (!(LAB & 16)) {
            ks.store(nimg_kv, tid);
            vs.store(nimg_kv + IMG, tid);
        }
        if constexpr (!(LAB & 2)) {
            const int jn = min(j + 2, ntile - 1);          // clamped: a static number of loads in flight keeps the counted waits exact
            ks.load(kbase + (size_t)jn * 32 * ldq, ldq, tid);
            vs.load(vbase + (size_t)jn * 32 * ldq, ldq, tid);
        }
        unsigned char* thr[QB];
#pragma unroll
        for (int X = 0; X < QB; ++X) {
            thr[X] = thT[X] + ql * 2 + a * (RPP * 64);
            if constexpr (P == 0) win_set<0>(T1[X].w, thr[X]);
            win_set<(P + 1) & 1>(T1[X].w, thr[X] + (P + 1) * 64);
        }
        const unsigned char* ei = eimg + P * EIMG;
        f32x16 sacc[QB], dpacc[QB];
        A4_FENCE();
        A4_STAMP(1);
        // ---- G1
        sacc[0] = A4_MFMA(ef0, as_frag(T0[0]), zero16());
        ktr[0][0] = A4_TR(kimg, la, 0, 0);
        A4_FENCE();
        dpacc[0] = A4_MFMA(vfr[0], dof[0][0], ndl[0]);
        ktr[0][1] = A4_TR(kimg, la, 0, 1);
        A4_FENCE();
        sacc[0] = A4_MFMA(ef1, as_frag(T1[0]), sacc[0]);
        ktr[1][0] = A4_TR(kimg, la, 1, 0);
        A4_FENCE();
        dpacc[0] = A4_MFMA(vfr[1], dof[0][1], dpacc[0]);
        ktr[1][1] = A4_TR(kimg, la, 1, 1);
        A4_FENCE();
        sacc[0] = A4_MFMA(kfr[0], qf[0][0], sacc[0]);
        etr[0] = A4_ETR(ei, ea, 0);
        A4_FENCE();
        dpacc[0] = A4_MFMA(vfr[2], dof[0][2], dpacc[0]);
        etr[1] = A4_ETR(ei, ea, 1);
        A4_FENCE();
        sacc[0] = A4_MFMA(kfr[1], qf[0][1], sacc[0]);
        dpacc[0] = A4_MFMA(vfr[3], dof[0][3], dpacc[0]);
        sacc[0] = A4_MFMA(kfr[2], qf[0][2], sacc[0]);
        sacc[0] = A4_MFMA(kfr[3], qf[0][3], sacc[0]);
        A4_FENCE();
        A4_STAMP(2);
        // ---- G2: block B's chains, block A's softmax arithmetic two elements per gap from the third MFMA on
        float ds[QB][16];
        auto valu2 = [&](int X, int r) {
            if constexpr (LAB & 1) {
                ds[X][r] = sacc[X][r] + dpacc[X][r];
                ds[X][r + 1] = sacc[X][r + 1] + dpacc[X][r + 1];
            } else {
                ds[X][r] = __builtin_amdgcn_exp2f(fmaf(sacc[X][r], sl, nlse2[X])) * dpacc[X][r];
                ds[X][r + 1] = __builtin_amdgcn_exp2f(fmaf(sacc[X][r + 1], sl, nlse2[X])) * dpacc[X][r + 1];
            }
        };
        sacc[1] = A4_MFMA(ef0, as_frag(T0[1]), zero16());
        dpacc[1] = A4_MFMA(vfr[0], dof[1][0], ndl[1]);
        A4_FENCE();
        sacc[1] = A4_MFMA(ef1, as_frag(T1[1]), sacc[1]);
        valu2(0, 0);
        A4_FENCE();
        dpacc[1] = A4_MFMA(vfr[1], dof[1][1], dpacc[1]);
        valu2(0, 2);
        A4_FENCE();
        sacc[1] = A4_MFMA(kfr[0], qf[1][0], sacc[1]);
        valu2(0, 4);
        A4_FENCE();
        dpacc[1] = A4_MFMA(vfr[2], dof[1][2], dpacc[1]);
        valu2(0, 6);
        A4_FENCE();
        sacc[1] = A4_MFMA(kfr[1], qf[1][1], sacc[1]);
        valu2(0, 8);
        A4_FENCE();
        dpacc[1] = A4_MFMA(vfr[3], dof[1][3], dpacc[1]);
        valu2(0, 10);
        A4_FENCE();
        sacc[1] = A4_MFMA(kfr[2], qf[1][2], sacc[1]);
        valu2(0, 12);
        A4_FENCE();
        sacc[1] = A4_MFMA(kfr[3], qf[1][3], sacc[1]);
        valu2(0, 14);
        A4_FENCE();
        const bf16x8 dsA0 = packfrag(ds[0]), dsA1 = packfrag(ds[0] + 8);
        A4_FENCE();
        A4_STAMP(3);
        // ---- G3: block A's second MFMA group, block B's softmax arithmetic in its gaps
        dq[0][0] = A4_MFMA(ktr[0][0], dsA0, dq[0][0]);
        valu2(1, 0);
        A4_FENCE();
        dq[0][1] = A4_MFMA(ktr[1][0], dsA0, dq[0][1]);
        valu2(1, 2);
        A4_FENCE();
        eacc[0] = A4_MFMA(etr[0], dsA0, eacc[0]);
        valu2(1, 4);
        A4_FENCE();
        dq[0][0] = A4_MFMA(ktr[0][1], dsA1, dq[0][0]);
        valu2(1, 6);
        A4_FENCE();
        dq[0][1] = A4_MFMA(ktr[1][1], dsA1, dq[0][1]);
        valu2(1, 8);
        A4_FENCE();
        eacc[0] = A4_MFMA(etr[1], dsA1, eacc[0]);
        valu2(1, 10);
        A4_FENCE();
        valu2(1, 12);
        valu2(1, 14);
        const bf16x8 dsB0 = packfrag(ds[1]), dsB1 = packfrag(ds[1] + 8);
        A4_FENCE();
        A4_STAMP(4);
        if constexpr (!(LAB & 4)) __syncthreads();
        A4_FENCE();
        A4_STAMP(5);
        // ---- G4: block B's second MFMA group, the next tile's row fragments in its gaps
        constexpr int PN = (P + 1) % PH;
        const unsigned char* ein = eimg + PN * EIMG;
        dq[1][0] = A4_MFMA(ktr[0][0], dsB0, dq[1][0]);
        ef0 = A4_EF(ein, ea, 0);
        ef1 = A4_EF(ein, ea, 1);
        A4_FENCE();
        dq[1][1] = A4_MFMA(ktr[1][0], dsB0, dq[1][1]);
        kfr[0] = A4_ROW(nimg_kv, la, 0);
        vfr[0] = A4_ROW(nimg_kv + IMG, la, 0);
        A4_FENCE();
        eacc[1] = A4_MFMA(etr[0], dsB0, eacc[1]);
        kfr[1] = A4_ROW(nimg_kv, la, 1);
        vfr[1] = A4_ROW(nimg_kv + IMG, la, 1);
        A4_FENCE();
        dq[1][0] = A4_MFMA(ktr[0][1], dsB1, dq[1][0]);
        kfr[2] = A4_ROW(nimg_kv, la, 2);
        vfr[2] = A4_ROW(nimg_kv + IMG, la, 2);
        A4_FENCE();
        dq[1][1] = A4_MFMA(ktr[1][1], dsB1, dq[1][1]);
        kfr[3] = A4_ROW(nimg_kv, la, 3);
        vfr[3] = A4_ROW(nimg_kv + IMG, la, 3);
        A4_FENCE();
        eacc[1] = A4_MFMA(etr[1], dsB1, eacc[1]);
        A4_FENCE();
        A4_STAMP(6);
        // key row 8 a + P is complete: its gradient (window slot P & 3 = a D row of half-wave 1) replaces the table entry; lanes of
        // half-wave 0 hold kw-gradient rows in these registers and write to their trash slot instead (no branch, no exec mask)
#pragma unroll
        for (int X = 0; X < QB; ++X) {
            {
                const float cur = eacc[X][win_reg(P & 3)];
                *reinterpret_cast<bf16*>(g ? thr[X] + P * 64 : trash) = (bf16)(cur - wprev[X][P & 3]);
                wprev[X][P & 3] = cur;
            }
            if constexpr (P == PH - 1) {
                const float cur = eacc[X][win_reg((P + 1) & 3)];
                *reinterpret_cast<bf16*>(g ? thr[X] + (P + 1) * 64 : trash) = (bf16)(cur - wprev[X][(P + 1) & 3]);
                wprev[X][(P + 1) & 3] = cur;
            }
        }
    };
    for (int a = 0; a < ((abl & 16) ? 0 : Hp / RPP); ++a) {
        body(std::integral_constant<int, 0>{}, a);
        body(std::integral_constant<int, 1>{}, a);
        body(std::integral_constant<int, 2>{}, a);
        body(std::integral_constant<int, 3>{}, a);
        body(std::integral_constant<int, 4>{}, a);
        body(std::integral_constant<int, 5>{}, a);
        body(std::integral_constant<int, 6>{}, a);
    }
    __syncthreads();          // the last iteration's window write-backs of every wave (own-wave data, but the r-space steps follow a common point)
#undef A4_FENCE
#undef A4_STAMP
#undef A4_TR
#undef A4_ETR
#undef A4_EF
#undef A4_ROW
#undef A4_MFMA

    // ---- r-space: dG[q][r] gathered from the two gradient tables; dQ^T += Rcat^T . dG^T; the dG fragments stay in registers.
    // The two blocks' chains (gather -> pack -> MFMA) are independent and run interleaved in one stream; the Rcat^T fragments are shared.
    uint4 gfs[QB][NSMAX];
    const int nkh = 2 * Hp - 1, nstep = NRP / 16;
    float* twgp[QB];
    int qhh[QB], qww[QB];
#pragma unroll
    for (int X = 0; X < QB; ++X) {
        twgp[X] = reinterpret_cast<float*>(smem + lds.twg + (wave * QB + X) * (32 * WP * 4));
        qhh[X] = q[X] / WP;
        qww[X] = q[X] % WP;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int rho = acc_row(reg, lane);
            if (rho < 22) twgp[X][ql * WP + rho] = eacc[X][reg];
            else if (rho >= 24 && rho < 30) twgp[X][ql * WP + rho - 2] = eacc[X][reg];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { dq[X][0][r] *= scale; dq[X][1][r] *= scale; }
    }
    // (three cases per step and half-wave, constant-offset reads, one range test per entry: attn3.hip)
    auto gather = [&](int X, int s, float (&gv)[8]) {
        const unsigned char* th = thT[X];
        const float* twg = twgp[X];
        const int qh = qhh[X], qw = qww[X];
        const int r0 = 16 * s + 8 * g;
        if (r0 + 7 < nkh) {
            const int kh0 = qh + Hp - 1 - r0;
            const unsigned char* bp = th + ql * 2 + (kh0 - 7) * 64;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const float vv = (float)*reinterpret_cast<const bf16*>(bp + (7 - t) * 64);
                gv[t] = (unsigned)(kh0 - t) < (unsigned)Hp ? vv : 0.f;
            }
        } else if (r0 >= nkh) {
            const int kw0 = qw + WP - 1 - (r0 - nkh);
            const float* bp = twg + ql * WP + (kw0 - 7);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const float vv = bp[7 - t];
                gv[t] = (unsigned)(kw0 - t) < (unsigned)WP ? vv : 0.f;
            }
        } else {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int r = r0 + t;
                const int khh = qh + Hp - 1 - r;
                const int kww = qw + WP - 1 - (r - nkh);
                const bool okh = r < nkh && (unsigned)khh < (unsigned)Hp;
                const bool okw = r >= nkh && (unsigned)kww < (unsigned)WP;
                const float vh = (float)*reinterpret_cast<const bf16*>(th + (okh ? khh : 0) * 64 + ql * 2);
                const float vw = twg[ql * WP + (okw ? kww : 0)];
                gv[t] = okh ? vh : (okw ? vw : 0.f);
            }
        }
    };
    {
        const unsigned char* r0p = rimg + ql * rpitch + 16 * g, *r1p = r0p + 32 * rpitch;
        auto rfrag = [&](const unsigned char* rp, int s) { return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(rp + 32 * s)); };
        bf16x8 rf[2] = {rfrag(r0p, 0), rfrag(r1p, 0)};
        float gcur[QB][8];
#pragma unroll
        for (int X = 0; X < QB; ++X) gather(X, 0, gcur[X]);
#pragma unroll
        for (int s = 0; s < NSMAX; ++s) {
#pragma unroll
            for (int X = 0; X < QB; ++X) gfs[X][s] = zero4();
            if (s < ((abl & 1) ? 0 : nstep)) {
                const int sn = min(s + 1, nstep - 1);
                const bf16x8 rn[2] = {rfrag(r0p, sn), rfrag(r1p, sn)};
                float gnext[QB][8];
#pragma unroll
                for (int X = 0; X < QB; ++X) gather(X, sn, gnext[X]);
#pragma unroll
                for (int X = 0; X < QB; ++X) {
                    const bf16x8 gf = packfrag(gcur[X]);
                    gfs[X][s] = __builtin_bit_cast(uint4, gf);
#pragma unroll
                    for (int db = 0; db < 2; ++db) dq[X][db] = mfma(rf[db], gf, dq[X][db]);
#pragma unroll
                    for (int t = 0; t < 8; ++t) gcur[X][t] = gnext[X][t];
                }
                rf[0] = rn[0];
                rf[1] = rn[1];
            }
        }
    }
#pragma unroll
    for (int X = 0; X < QB; ++X) {
        unsigned char* stg = smem + lds.stg + (wave * QB + X) * IMG;
        stage_rows(stg, dq[X], 1.f, lane);
        write_rows(stg, dqkv + (size_t)(b * L + qt[X] * 32) * ldq + h * ATT_HD, ldq, lane);      // same-wave LDS operations are ordered
    }

    // ---- d Rcat^T[d][r] += Q^T[d][q] dG^T[r][q] over the workgroup's 256 queries (attn3.hip FUSE, with eight source blocks)
    __syncthreads();
    const int nimg = (NRP + 63) >> 6;
    {
        const int rsw = vsw(ql);
#pragma unroll
        for (int X = 0; X < QB; ++X) {
            unsigned char* wimg = smem + (wave * QB + X) * (nimg + 1) * IMG;
#pragma unroll
            for (int s = 0; s < NSMAX; ++s)
                if (s < nstep) *reinterpret_cast<uint4*>(wimg + (s >> 2) * IMG + ql * 128 + ((((2 * s + g) & 7) ^ rsw) << 4)) = gfs[X][s];
#pragma unroll
            for (int s = 0; s < 4; ++s)
                *reinterpret_cast<uint4*>(wimg + nimg * IMG + ql * 128 + (((2 * s + g) ^ rsw) << 4)) = __builtin_bit_cast(uint4, qf[X][s]);
        }
    }
    __syncthreads();
    // 12 units (r-block, d-block) of one 32 x 32 accumulator over the 8 source blocks; wave w takes units w, w + 4, w + 8
    float* pw = part + (size_t)blockIdx.x * NRP * ATT_HD;
    for (int u = wave; u < ((abl & 2) ? 0 : 2 * (NRP / 32)); u += NW) {
        const int rb = u >> 1, db = u & 1;
        f32x16 acc = zero16();
#pragma unroll
        for (int w2 = 0; w2 < WGB; ++w2) {
            const unsigned char* im = smem + w2 * (nimg + 1) * IMG;
#pragma unroll
            for (int ksx = 0; ksx < 2; ++ksx)
                acc = mfma(trfrag(im + nimg * IMG, la, db, ksx), trfrag(im + (rb >> 1) * IMG, la, rb & 1, ksx), acc);
        }
        float* prow = pw + (size_t)(rb * 32 + ql) * ATT_HD + 4 * g + db * 32;                      // D: lane = r, registers = d
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
            *reinterpret_cast<float4*>(prow + 8 * rg) = make_float4(acc[rg * 4], acc[rg * 4 + 1], acc[rg * 4 + 2], acc[rg * 4 + 3]);
    }
}

}   // namespace a4

// ---------------------------------------------------------------------------------------------- host side
// OFF by default (round-4 measurements in the header / DESIGN.md section 4.5: it does not beat generation 3 yet); PA_ATTN4=1 or
// pa_debug_set(9, 2) turns it on (the parity test does)
static int a4_on() {
    static const int v = [] { const char* e = getenv("PA_ATTN4"); return e ? atoi(e) : 0; }();
    return g_attn4 == 1 ? 0 : (g_attn4 == 2 ? 1 : v);
}
// whole 8-tile groups per head the generation-4 kernels take (0: generation 4 not used for this grid)
int attn4_groups(int L, int Hp, int Wp) {
    if (!a4_on() || !attn3_ok(L, Hp, Wp)) return 0;
    if (pa_relpos_rows_padded(Hp, Wp) > 16 * a4::NSMAX) return 0;
    return (L / 32) / a4::WGB;
}
int attn4_bwd_dq(const bf16* qkv, int64_t ldq, const bf16* rcatT, const bf16* dout, int64_t lddo, const float* lse, const void* tables,
                 bf16* dqkv, float* part, int Bn, int L, int H, int Hp, int Wp, float scale, int xcd_map, hipStream_t st) {
    using namespace a4;
    const int NRP = pa_relpos_rows_padded(Hp, Wp);
    const int ngrp = attn4_groups(L, Hp, Wp);
    if (ngrp <= 0 || part == nullptr) return (int)hipErrorInvalidValue;
    const DqLds lds(Hp, NRP);
    static const int lab = [] { const char* v = getenv("PA_ATTN4_DQ_LAB"); return v ? atoi(v) : 0; }();
    auto kern = bwd_dq64_kernel<0>;
#ifdef A4_ABLATION_BUILD
    switch (lab) {
    case 1: kern = bwd_dq64_kernel<1>; break;
    case 2: kern = bwd_dq64_kernel<2>; break;
    case 4: kern = bwd_dq64_kernel<4>; break;
    case 8: kern = bwd_dq64_kernel<8>; break;
    case 16: kern = bwd_dq64_kernel<16>; break;
    case 18: kern = bwd_dq64_kernel<18>; break;
    case 32: kern = bwd_dq64_kernel<32>; break;
    case 33: kern = bwd_dq64_kernel<33>; break;
    case 63: kern = bwd_dq64_kernel<63>; break;
    case 64: kern = bwd_dq64_kernel<64>; break;
    }
#else
    (void)lab;
#endif
    static bool done = false;
    if (int e = set_smem(reinterpret_cast<const void*>(kern), done)) return e;
    PA_LAUNCH(kern, dim3(ngrp * Bn * H), dim3(NT), (size_t)lds.total, st, qkv, (size_t)ldq, rcatT, dout, (size_t)lddo, lse,
              reinterpret_cast<const unsigned char*>(tables), dqkv, part, L, H, Hp, NRP, scale, ngrp, xcd_map,
              [] { const char* v = getenv("PA_ATTN4_DQ_ABL"); return v ? atoi(v) : 0; }());
    return (int)hipGetLastError();
}

// the 64-key dKV kernel: not built yet -- generation 3 keeps every key tile
bool attn4_dkv_on() { return false; }
int attn4_bwd_dkv(const bf16*, int64_t, const bf16*, int64_t, const void*, bf16*, int, int, int, int, int, float, int, hipStream_t) {
    return (int)hipErrorInvalidValue;
}

// diagnostics (ablation build, PA_ATTN4_DQ_LAB=64): s_memtime stamps of workgroup 0, thread 0: 64 iterations x 8 points
extern "C" int pa_attn4_trace(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(a4::g_trace4), sizeof(a4::g_trace4));
}
