// bf16 fused attention with decomposed rel-pos bias, third generation: forward, backward-dQ, backward-dKV for token grids whose key
// rows are 28 tokens wide (the 896x448 / patch-16 grid of every reference factory: Painter/models_painter.py:476-487,
// SegGPT_inference/models_seggpt.py).  Same math and C ABI as attn2.hip (Painter/models_painter.py:76-86, util/vitdet_utils.py:63-125,
// SURVEY.md 8a a5-a8, a17, Appendix B.2); attn2.hip keeps the other grids, attn_fwd.hip / attn_bwd.hip the exact-fp32 build.
//
// Why a third generation.  PMC of attn2 (round 1): the VALU, not the matrix pipe, is the busy unit (per 32x32 tile and wave ~700 VALU
// cycles against 256-512 MFMA cycles) and waves sit ~45 % of their cycles in s_waitcnt/barrier.  A good part of the VALU stream and
// the dependent LDS reads (run table -> kw table -> kh table) only served the rel-pos bias.  Here the bias costs no VALU at all:
//
//   S[q][key] = q.k + sum_j T[q][j] E[key][j]          one extra 32-deep contraction = 2 MFMAs per 32x32 tile
//
//   T[q][j]  (bf16, per query):   28 entries  tw[q][kw]  = (q . rel_pos_w[qw - kw + 27]) / scale
//                                  4 entries  "window"   = (q . rel_pos_h[qh - kh + Hp-1]) / scale  of the <= 2 key rows a tile touches
//   E[key][j] (one-hot, bf16):    1 at j = kw(key) and at the window slot (kh(key) & 3)
//
// 7 key tiles of 32 = 8 key rows of 28: the one-hot patterns repeat with period 7 (7 images of 2 KB in LDS), the loop is unrolled over
// the 7 phases so every offset is an immediate, tile p of a period touches key rows p and p+1, and a key row r lives in window slot
// r & 3.  The four window slots sit in ONE VGPR pair position of the T operand (slots (t = 6, 7) of both half-waves), so moving the
// window forward is one 16-bit LDS read per tile, for all lanes alike.  The bias GRADIENT is the transposed contraction
// dT[j][q] += E^T[j][key] dS^T[key][q] (2 MFMAs, the same LDS image read with the transposing ds_read_b64_tr_b16): rows of kw
// accumulate over all tiles, a window row is complete after the tile that bears its number and is written back as bf16 into the kh
// table entry it replaces.  In dKV (lane = key, registers = queries) the row log-sum-exp rides in the two window slots the wave's two
// key rows leave free, as a bf16 hi + lo pair against E = 1, so S - lse also comes out of the matrix pipe.
// Per 32x32 tile and wave: forward 10 MFMAs / ~75 VALU (was 8 / ~115), dQ 16 / ~65 (14 / ~130), dKV 18 / ~70 (16 / ~100).
//
// Table tiles.  The forward builds T once per query (G = Rcat . Q^T on the matrix pipe, as before) and, when a backward will follow,
// writes it to `tables`: per (sample, head, 32-query tile) a block of
//     [32 q][32 slots] bf16 kw part (window slots zero) | [Hp + 2][32 q] bf16 kh part, transposed; rows Hp, Hp+1 = -lse/scale hi, lo
//     | [32 q] f32 -Delta                                                          (rounded up to 256 B; 6 KB at Hp = 56)
// The backward's prep kernel fills the lse rows and -Delta; dQ reads its own rows once, dKV streams the kw part and the <= 8 kh rows
// its workgroup needs with every query tile (10.7 KB per tile instead of attn2's 19 KB).  Because forward and backward contract the
// very same bf16 T entries, P is recomputed in the backward from exactly the logits the forward saw.
#include "attn3_common.h"
#include "../../include/painter_hip.h"
#include "attn3.h"
#include <cstdlib>

#ifndef A3_PRIO
#define A3_PRIO 0
#endif
#ifndef A3_HOIST
#define A3_HOIST 1      // 0: no sched_barriers around the hoisted fragment loads of the backward kernels (experiment)
#endif
#ifndef A3_NDL
#define A3_NDL true    // false: -Delta added on the VALU instead of riding in the dP accumulator init (experiment)
#endif
// A3_PRIO = 1: raise the wave priority while it issues an MFMA group (experiment)
#define A3_PRIO_UP() do { if (A3_PRIO) __builtin_amdgcn_s_setprio(1); } while (0)
#define A3_PRIO_DOWN() do { if (A3_PRIO) __builtin_amdgcn_s_setprio(0); } while (0)

namespace a3 {

// =============================================================================================== forward
// LDS: [K img | V img] x STAGES | thT 4 waves x [Hp][32 q] bf16 | 7 one-hot images (the per-wave T rows alias them during the prologue)
template <int STAGES>
__global__ __launch_bounds__(NT, STAGES == 1 ? 4 : 2) void fwd_kernel(const bf16* __restrict__ qkv, size_t ldq, const bf16* __restrict__ rcat,
                                                                      bf16* __restrict__ out, size_t ldo, float* __restrict__ lse,
                                                                      unsigned char* __restrict__ tables, int L, int H, int Hp, int NRP,
                                                                      float scale, int nblk, int xcd_map, int abl) {
    // abl (diagnostics, PA_ATTN3_FWD_ABL; results WRONG when set): 16 no key loop, 32 no table build in the prologue
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 5, ql = lane & 31;
    int blk, bh;
    wg_coords(nblk, xcd_map, blk, bh, (L / 32) / NW);
    const int b = bh / H, h = bh % H, D = H * ATT_HD;
    const bf16* base = qkv + (size_t)b * L * ldq + h * ATT_HD;
    const bf16* kbase = base + D;
    const bf16* vbase = base + 2 * D;
    const int qt = blk * NW + wave;
    const bool valid = qt * 32 < L;
    const int q = qt * 32 + ql;
    unsigned char* thT = smem + STAGES * STAGE_QK + wave * Hp * 64;
    unsigned char* eimg = smem + STAGES * STAGE_QK + NW * Hp * 64;
    unsigned char* twimg = eimg + wave * 2048;
    LaneAddr la;
    la.init(lane);
    EAddr ea;
    ea.init(lane);
    const int ntile = L / 32;
    Stager ks, vs;
    ks.load(kbase, ldq, tid);
    vs.load(vbase, ldq, tid);

    bf16x8 qf[4];
    uint4 T0 = zero4(), T1 = zero4();
    if (valid) {
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = gfrag(base + (size_t)q * ldq, s, g);
        if (!(abl & 32)) build_tables3(twimg, thT, rcat, NRP, qf, q / WP, q % WP, Hp, 1.f / scale, lane, smem + tid * 2);      // trash: the K/V stage, still unused
        T0 = *reinterpret_cast<const uint4*>(twimg + ql * 64 + 16 * g);          // same-wave LDS ops are ordered
        T1 = *reinterpret_cast<const uint4*>(twimg + ql * 64 + 32 + 16 * g);
        if (tables != nullptr) {
            unsigned char* tt = tables + ((size_t)bh * ntile + qt) * ttile_bytes(Hp);
#pragma unroll
            for (int i = 0; i < 2; ++i) *reinterpret_cast<uint4*>(tt + (lane + 64 * i) * 16) = *reinterpret_cast<const uint4*>(twimg + (lane + 64 * i) * 16);
            for (int c = lane; c < Hp * 4; c += 64) *reinterpret_cast<uint4*>(tt + 2048 + c * 16) = *reinterpret_cast<const uint4*>(thT + c * 16);
        }
    }
    __syncthreads();                    // the T rows in the image region are dead
    build_eimg(eimg, tid);
    ks.store(smem, tid);
    vs.store(smem + IMG, tid);
    __syncthreads();

    f32x16 oacc[2];
    oacc[0] = zero16();
    oacc[1] = zero16();
    float m = 0.f, l = 0.f;
    const float sl = scale * LOG2E_F;
    const unsigned char* thw = thT + ql * 2;

    auto body = [&](auto pc, int a) {
        constexpr int P = decltype(pc)::value;
        const int j = a * PH + P;
        {   // unconditional (clamped to the last tile): a static number of loads in flight
            const int jn = min(j + 1, ntile - 1);
            ks.load(kbase + (size_t)jn * 32 * ldq, ldq, tid);
            vs.load(vbase + (size_t)jn * 32 * ldq, ldq, tid);
        }
        const unsigned char* kimg = smem + (STAGES == 2 ? (j & 1) * STAGE_QK : 0);
        const unsigned char* vimg = kimg + IMG;
        if (valid) {
            const unsigned char* thr = thw + a * (RPP * 64);
            if constexpr (P == 0) win_set<0>(T1.w, thr);
            win_set<(P + 1) & 1>(T1.w, thr + (P + 1) * 64);
            const unsigned char* ei = eimg + P * EIMG;
            // every operand fragment of the S chain is requested before the first MFMA (the scheduler otherwise alternates
            // read - wait - MFMA through one register quad and pays the LDS latency six times per tile)
            const bf16x8 ef0 = efrag(ei, ea, 0), ef1 = efrag(ei, ea, 1);
            bf16x8 kfr[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) kfr[s] = rowfrag(kimg, la, s);
            __builtin_amdgcn_sched_barrier(0);
            f32x16 sacc = zero16();
            A3_PRIO_UP();
            sacc = mfma(ef0, as_frag(T0), sacc);
            sacc = mfma(ef1, as_frag(T1), sacc);
#pragma unroll
            for (int s = 0; s < 4; ++s) sacc = mfma(kfr[s], qf[s], sacc);
            A3_PRIO_DOWN();
            // V fragments travel while the softmax runs on the VALU (the 128-register single-stage build has room for half of them)
            bf16x8 vtr[2][2];
            vtr[0][0] = trfrag(vimg, la, 0, 0);
            vtr[0][1] = trfrag(vimg, la, 0, 1);
            if constexpr (STAGES == 2) { vtr[1][0] = trfrag(vimg, la, 1, 0); vtr[1][1] = trfrag(vimg, la, 1, 1); }
            __builtin_amdgcn_sched_barrier(0);
            float p[16];
            const float nm = -m;
#pragma unroll
            for (int r = 0; r < 16; ++r) p[r] = fmaf(sacc[r], sl, nm);
            float tmax = max16(p);
            tmax = fmaxf(tmax, xor32(tmax));
            if (j == 0 || __any(tmax > THR)) {          // wave-uniform; after the first tiles almost never taken
                const float delta = (j == 0) ? tmax : fmaxf(tmax, 0.f);
                const float alpha = __builtin_amdgcn_exp2f(-delta);
                m += delta;
                l *= alpha;
#pragma unroll
                for (int r = 0; r < 16; ++r) { oacc[0][r] *= alpha; oacc[1][r] *= alpha; p[r] -= delta; }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) p[r] = __builtin_amdgcn_exp2f(p[r]);
            l += sum16(p);
            const bf16x8 pf0 = packfrag(p), pf1 = packfrag(p + 8);
            if constexpr (STAGES == 1) { vtr[1][0] = trfrag(vimg, la, 1, 0); vtr[1][1] = trfrag(vimg, la, 1, 1); }
            A3_PRIO_UP();
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                oacc[db] = mfma(vtr[db][0], pf0, oacc[db]);
                oacc[db] = mfma(vtr[db][1], pf1, oacc[db]);
            }
            A3_PRIO_DOWN();
        }
        if constexpr (STAGES == 1) __syncthreads();      // every wave has finished reading the only stage
        if (j + 1 < ntile) {
            ks.store(smem + (STAGES == 2 ? ((j + 1) & 1) * STAGE_QK : 0), tid);
            vs.store(smem + (STAGES == 2 ? ((j + 1) & 1) * STAGE_QK : 0) + IMG, tid);
        }
        __syncthreads();
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
    // per-wave 4 KB staging tile: the K/V stages (2 stages) or, with the single stage, the one-hot images -- both are free now
    unsigned char* stg = smem + wave * IMG;
    if constexpr (STAGES == 1) { if (wave >= 2) stg = eimg + (wave - 2) * IMG; }
    if (valid) {
        const float lt = l + xor32(l);
        if (g == 0) {
            const float lse_q = (m + __builtin_amdgcn_logf(lt)) * LN2_F;
            lse[(size_t)bh * L + q] = lse_q;
            if (tables != nullptr) {      // -lse / scale as the bf16 hi + lo pair the dKV kernel contracts (rows Hp, Hp + 1 of the kh part): round 5,
                unsigned char* tt = tables + ((size_t)bh * ntile + qt) * ttile_bytes(Hp);      // was a separate launch in front of the backward
                const float x = -lse_q * (1.f / scale);      // the same multiply the prep kernels use: the "fused" and "launch" routes stay bit-comparable for any scale
                const bf16 hi = (bf16)x;
                *reinterpret_cast<bf16*>(tt + 2048 + Hp * 64 + ql * 2) = hi;
                *reinterpret_cast<bf16*>(tt + 2048 + (Hp + 1) * 64 + ql * 2) = (bf16)(x - (float)hi);
            }
        }
        stage_rows(stg, oacc, 1.f / lt, lane);
        write_rows(stg, out + (size_t)(b * L + qt * 32) * ldo + h * ATT_HD, ldo, lane);   // same-wave LDS ops are ordered
    }
}

// =============================================================================================== backward: dQ, bias gradients
// LDS: [K img | V img] x 2 | thT 4 waves x [Hp][32 q] bf16 (values, replaced row by row by their gradients) | 7 one-hot images
// NDL = true keeps -Delta in 16 accumulator-init registers (dP - Delta comes out of the MFMA chain); false adds it on the VALU
// diagnostics (pa_attn_trace): s_memtime stamps of two workgroups' waves 0 / 1 at seven points of every tile iteration
__device__ unsigned long long g_trace[2 * 2 * 64 * 8];
// FUSE: the rel-pos table gradient d Rcat[r][d] = sum_q dG[q][r] Q[q][d] is contracted here, per workgroup, instead of writing the
// per-query bias gradient dG (bf16 [R, heads * NRP], 77 MB per launch at the ViT-L shape) for a gather GEMM over it: after the r-space
// loop the waves put their dG rows (bf16 [32 q][NRP], kept in registers through that loop) and their Q rows into LDS images -- every
// other LDS region is dead by then -- and wave w contracts r-blocks w, w + 4 over all of the workgroup's queries through the
// transposing reads (d Rcat^T[d][r] += Q^T[d][q] dG^T[r][q]); one fp32 [NRP][64] partial per workgroup goes to `part`
// (pa_attn_bwd_relpos_reduce sums them in a fixed order).  Needs NRP <= 16 * NSMAX.
constexpr int NSMAX = 12;
template <int MINW, bool NDL, bool TR = false, bool FUSE = false>
__global__ __launch_bounds__(NT, MINW) void bwd_dq_kernel(const bf16* __restrict__ qkv, size_t ldq, const bf16* __restrict__ rcatT,
                                                          const bf16* __restrict__ dout, size_t lddo, const float* __restrict__ lse,
                                                          unsigned char* __restrict__ tables, bf16* __restrict__ dqkv,
                                                          bf16* __restrict__ dG, float* __restrict__ part, int L, int H, int Hp, int NRP,
                                                          float scale, int nblk, int xcd_map, int abl, int tile0, int pslot0,
                                                          const bf16* __restrict__ oatt, size_t ldo) {
    // oatt (round 5, may be NULL): the forward's output O.  Given, this kernel computes Delta = rowsum(dO o O) of its own query rows in the
    // prologue (each lane holds half of its row's dO: 32 products + one half-wave exchange) and writes -Delta into the table tile for the
    // dKV launch behind it -- the prep launch that did this for every block (24 x 15 us per step on the main stream) is gone; the lse
    // fields of the tile come from the forward.  NULL: -Delta is read from the tile (pa_attn_bwd_prep or the prep kernel wrote it).
    // tile0: first 32-query tile of every head this launch covers (0 in the product; a launch may cover the tail of every head only);
    // pslot0: first partial slot of this launch in `part`
    // abl (diagnostics, PA_ATTN3_DQ_ABL; results are WRONG with any bit set): 1 no r-space loop after the key loop, 2 no dG stores,
    // 4 no dQ store, 16 no key loop, 32 no gather in the r-space loop, 64 no MFMA in the r-space loop
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 5, ql = lane & 31;
    const int trwg = blockIdx.x == 0 ? 0 : (blockIdx.x == gridDim.x / 2 ? 1 : -1);
    auto coarse = [&](int pt) {          // traced build: stamps 60 kernel start, 61 loop start, 62 loop end, 63 kernel end (slot 0 of each)
        if constexpr (TR) {
            if constexpr (MINW < 3 && A3_HOIST) __builtin_amdgcn_sched_barrier(0);      // the 3-wave build leaves the order to the scheduler (register budget 168)
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (trwg >= 0 && wave < 2 && lane == 0) g_trace[((trwg * 2 + wave) * 64 + pt) * 8] = t;
            if constexpr (MINW < 3 && A3_HOIST) __builtin_amdgcn_sched_barrier(0);      // the 3-wave build leaves the order to the scheduler (register budget 168)
        }
    };
    coarse(60);
    int blk, bh;
    wg_coords(nblk, xcd_map, blk, bh, (L / 32 - tile0) / NW);
    const int b = bh / H, h = bh % H, D = H * ATT_HD;
    const bf16* base = qkv + (size_t)b * L * ldq + h * ATT_HD;
    const bf16* kbase = base + D;
    const bf16* vbase = base + 2 * D;
    const int qt = tile0 + blk * NW + wave;
    const bool valid = qt * 32 < L;
    const int q = qt * 32 + ql;
    const int qh = q / WP, qw = q % WP;
    unsigned char* thT = smem + 2 * STAGE_QK + wave * Hp * 64;
    unsigned char* eimg = smem + 2 * STAGE_QK + NW * Hp * 64;
    // Rcat^T [64 d][NRP] bf16, the A operand of the r-space step after the key loop, staged once per workgroup: fetching its fragments
    // from global memory step by step made that short loop latency-bound (11 dependent L2 round trips; ablation: 33 us of the 213 us
    // kernel at the ViT-L grid).  Row pitch NRP * 2 + 16 bytes keeps the 16-byte fragment reads of 32 rows off each other's banks.
    unsigned char* rimg = eimg + PH * EIMG;
    const int rpitch = NRP * 2 + 16;
    LaneAddr la;
    la.init(lane);
    EAddr ea;
    ea.init(lane);
    const int ntile = L / 32;
    const float sl = scale * LOG2E_F;
    Stager ks, vs;
    ks.load(kbase, ldq, tid);
    vs.load(vbase, ldq, tid);
    const int rchunks = ATT_HD * (NRP / 8);               // 16-byte chunks of Rcat^T

    // every global load of the prologue is in flight before the LDS work (one-hot images, table copy) starts
    bf16x8 qf[4], dof[4];
    uint4 T0 = zero4(), T1 = zero4();
    uint4 tch[4] = {zero4(), zero4(), zero4(), zero4()};          // this lane's chunks of the kh table (Hp * 4 chunks of 16 B per wave)
    float nlse2 = 0.f, ndlt = 0.f;
    unsigned char* tt = tables + ((size_t)bh * ntile + qt) * ttile_bytes(Hp);
    if (valid) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            qf[s] = gfrag(base + (size_t)q * ldq, s, g);
            dof[s] = gfrag(dout + (size_t)(b * L + q) * lddo + h * ATT_HD, s, g);
        }
        if (oatt != nullptr) {
            float dl = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bf16x8 of = gfrag(oatt + (size_t)(b * L + q) * ldo + h * ATT_HD, s, g);
#pragma unroll
                for (int t = 0; t < 8; ++t) dl = fmaf((float)of[t], (float)dof[s][t], dl);
            }
            dl += xor32(dl);                                   // the other half of the row's 64 products
            ndlt = -dl;
            if (g == 0) *reinterpret_cast<float*>(tt + 2048 + (Hp + 2) * 64 + ql * 4) = ndlt;
        }
        T0 = *reinterpret_cast<const uint4*>(tt + ql * 64 + 16 * g);
        T1 = *reinterpret_cast<const uint4*>(tt + ql * 64 + 32 + 16 * g);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (lane + 64 * i < Hp * 4) tch[i] = *reinterpret_cast<const uint4*>(tt + 2048 + (lane + 64 * i) * 16);
        nlse2 = -lse[(size_t)bh * L + q] * LOG2E_F;
        if (oatt == nullptr) ndlt = *reinterpret_cast<const float*>(tt + 2048 + (Hp + 2) * 64 + ql * 4);
    }
    build_eimg(eimg, tid);
    {   // 22 KB from L2, two chunks in flight per thread (held in registers across the prologue it made the kernel spill)
        const int per_row = NRP / 8;
        for (int c = tid; c < rchunks; c += 2 * NT) {
            const int c2 = c + NT;
            const uint4 v0 = *reinterpret_cast<const uint4*>(rcatT + (size_t)c * 8);
            const uint4 v1 = c2 < rchunks ? *reinterpret_cast<const uint4*>(rcatT + (size_t)c2 * 8) : zero4();
            *reinterpret_cast<uint4*>(rimg + (c / per_row) * rpitch + (c % per_row) * 16) = v0;
            if (c2 < rchunks) *reinterpret_cast<uint4*>(rimg + (c2 / per_row) * rpitch + (c2 % per_row) * 16) = v1;
        }
    }
    if (valid) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (lane + 64 * i < Hp * 4) *reinterpret_cast<uint4*>(thT + (lane + 64 * i) * 16) = tch[i];
        for (int c = lane + 256; c < Hp * 4; c += 64) *reinterpret_cast<uint4*>(thT + c * 16) = *reinterpret_cast<const uint4*>(tt + 2048 + c * 16);
    }
    f32x16 ndl;
#pragma unroll
    for (int r = 0; r < 16; ++r) ndl[r] = NDL ? ndlt : 0.f;
    ks.store(smem, tid);
    vs.store(smem + IMG, tid);
    __syncthreads();

    f32x16 dq[2], eacc;
    dq[0] = zero16();
    dq[1] = zero16();
    eacc = zero16();
    unsigned char* thw = thT + ql * 2;

    coarse(61);
    auto body = [&](auto pc, int a) {
        constexpr int P = decltype(pc)::value;
        const int j = a * PH + P;
        auto mark = [&](int pt) {
            if constexpr (TR) {
                if constexpr (MINW < 3 && A3_HOIST) __builtin_amdgcn_sched_barrier(0);      // the 3-wave build leaves the order to the scheduler (register budget 168)
                const unsigned long long t = __builtin_amdgcn_s_memtime();
                if (trwg >= 0 && wave < 2 && lane == 0 && j < 60) g_trace[((trwg * 2 + wave) * 64 + j) * 8 + pt] = t;
                if constexpr (MINW < 3 && A3_HOIST) __builtin_amdgcn_sched_barrier(0);      // the 3-wave build leaves the order to the scheduler (register budget 168)
            }
        };
        mark(0);
        if (j + 1 < ntile) {
            ks.load(kbase + (size_t)(j + 1) * 32 * ldq, ldq, tid);
            vs.load(vbase + (size_t)(j + 1) * 32 * ldq, ldq, tid);
        }
        const unsigned char* kimg = smem + (j & 1) * STAGE_QK;
        const unsigned char* vimg = kimg + IMG;
        if (valid) {
            unsigned char* thr = thw + a * (RPP * 64);
            if constexpr (P == 0) win_set<0>(T1.w, thr);
            win_set<(P + 1) & 1>(T1.w, thr + (P + 1) * 64);
            const unsigned char* ei = eimg + P * EIMG;
            // all ten operand fragments of the S and dP chains are requested before the first MFMA
            const bf16x8 ef0 = efrag(ei, ea, 0), ef1 = efrag(ei, ea, 1);
            bf16x8 kfr[4], vfr[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) { vfr[s] = rowfrag(vimg, la, s); kfr[s] = rowfrag(kimg, la, s); }
            if constexpr (MINW < 3 && A3_HOIST) __builtin_amdgcn_sched_barrier(0);      // the 3-wave build leaves the order to the scheduler (register budget 168)
            if constexpr (TR) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
            mark(1);
            f32x16 sacc = zero16(), dpacc;
            A3_PRIO_UP();
            sacc = mfma(ef0, as_frag(T0), sacc);
            dpacc = mfma(vfr[0], dof[0], ndl);
            sacc = mfma(ef1, as_frag(T1), sacc);
#pragma unroll
            for (int s = 1; s < 4; ++s) dpacc = mfma(vfr[s], dof[s], dpacc);
#pragma unroll
            for (int s = 0; s < 4; ++s) sacc = mfma(kfr[s], qf[s], sacc);
            A3_PRIO_DOWN();
            // the transposed fragments of the second MFMA group travel while the VALU turns S, dP into dS
            bf16x8 ktr[2][2], etr[2];
#pragma unroll
            for (int db = 0; db < 2; ++db) { ktr[db][0] = trfrag(kimg, la, db, 0); ktr[db][1] = trfrag(kimg, la, db, 1); }
            etr[0] = etrfrag(ei, ea, 0);
            etr[1] = etrfrag(ei, ea, 1);
            if constexpr (MINW < 3 && A3_HOIST) __builtin_amdgcn_sched_barrier(0);      // the 3-wave build leaves the order to the scheduler (register budget 168)
            mark(2);
            float ds[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pr = __builtin_amdgcn_exp2f(fmaf(sacc[r], sl, nlse2));
                ds[r] = NDL ? pr * dpacc[r] : pr * (dpacc[r] + ndlt);
            }
            const bf16x8 dsf0 = packfrag(ds), dsf1 = packfrag(ds + 8);
            mark(3);
            A3_PRIO_UP();
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                dq[db] = mfma(ktr[db][0], dsf0, dq[db]);
                dq[db] = mfma(ktr[db][1], dsf1, dq[db]);
            }
            eacc = mfma(etr[0], dsf0, eacc);
            eacc = mfma(etr[1], dsf1, eacc);
            A3_PRIO_DOWN();
            // key row 8 a + P is complete: its gradient (window slot P & 3 = a D row of half-wave 1) replaces the table entry
            if (g) {
                *reinterpret_cast<bf16*>(thr + P * 64) = (bf16)eacc[win_reg(P & 3)];
                eacc[win_reg(P & 3)] = 0.f;
                if constexpr (P == PH - 1) {
                    *reinterpret_cast<bf16*>(thr + (P + 1) * 64) = (bf16)eacc[win_reg((P + 1) & 3)];
                    eacc[win_reg((P + 1) & 3)] = 0.f;
                }
            }
        }
        mark(4);
        if (j + 1 < ntile) {
            ks.store(smem + ((j + 1) & 1) * STAGE_QK, tid);
            vs.store(smem + ((j + 1) & 1) * STAGE_QK + IMG, tid);
        }
        mark(5);
        __syncthreads();
        mark(6);
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
    coarse(62);
    // every wave is past its last read of the K/V stages and the one-hot images (the loop ends with a barrier): the image region
    // becomes the fp32 kw-gradient table [wave][32 q][28], the K/V region the per-wave dQ staging tiles
    float* twg = reinterpret_cast<float*>(eimg + wave * (32 * WP * 4));
    unsigned char* stg = smem + wave * IMG;
    uint4 gfs[FUSE ? NSMAX : 1];
    if constexpr (FUSE) {
#pragma unroll
        for (int s = 0; s < NSMAX; ++s) gfs[s] = zero4();
    }
    if (valid) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int rho = acc_row(reg, lane);
            if (rho < 22) twg[ql * WP + rho] = eacc[reg];
            else if (rho >= 24 && rho < 30) twg[ql * WP + rho - 2] = eacc[reg];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { dq[0][r] *= scale; dq[1][r] *= scale; }
        // r-space: dG[q][r] gathers the tables; dQ^T[d][q] += sum_r Rcat[r][d] dG[q][r]; dG is also the operand of d rel_pos.
        // (the table gradients are plain sums of dS over the keys of a kw / kh class = d loss / d G: the bias enters the logit with
        // coefficient 1, the 1 / scale inside T and the scale inside `sl` cancel)
        bf16* dgrow = dG + ((size_t)(b * L + q) * H + h) * NRP;
        // the Rcat^T fragments of step s+1 are requested before the gather of step s (the loop is latency-bound otherwise: one L2 round
        // trip per step, ~1/4 of a workgroup's life at the ViT-L grid)
        // The gather is branch-free (clamped addresses, selects) and one step ahead of its use, like the fragment loads: with a branch per
        // element the eleven steps cost ~2000 cycles each (coarse stamps: 22k of a workgroup's 150k cycles sat in this loop).
        const int nkh = 2 * Hp - 1;
        // Three cases per step and half-wave: eight rel_pos_h entries, eight rel_pos_w entries, or the one step that straddles the two.
        // The pure cases read unconditionally at base + constant offsets (an out-of-range entry is some other halfword of this
        // workgroup's LDS and is selected away by ONE unsigned range test): ~5 VALU per element instead of ~12.
        auto gather = [&](int s, float (&gv)[8]) {
            const int r0 = 16 * s + 8 * g;
            if (r0 + 7 < nkh) {
                const int kh0 = qh + Hp - 1 - r0;                                  // entry t: kh0 - t
                const unsigned char* base = thT + ql * 2 + (kh0 - 7) * 64;
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const float v = (float)*reinterpret_cast<const bf16*>(base + (7 - t) * 64);
                    gv[t] = (unsigned)(kh0 - t) < (unsigned)Hp ? v : 0.f;
                }
            } else if (r0 >= nkh) {
                const int kw0 = qw + WP - 1 - (r0 - nkh);                          // entry t: kw0 - t (negative for the padding rows)
                const float* base = twg + ql * WP + (kw0 - 7);
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const float v = base[7 - t];
                    gv[t] = (unsigned)(kw0 - t) < (unsigned)WP ? v : 0.f;
                }
            } else {
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const int r = r0 + t;
                    const int khh = qh + Hp - 1 - r;
                    const int kww = qw + WP - 1 - (r - nkh);
                    const bool okh = r < nkh && (unsigned)khh < (unsigned)Hp;
                    const bool okw = r >= nkh && (unsigned)kww < (unsigned)WP;
                    const float vh = (float)*reinterpret_cast<const bf16*>(thT + (okh ? khh : 0) * 64 + ql * 2);
                    const float vw = twg[ql * WP + (okw ? kww : 0)];
                    gv[t] = okh ? vh : (okw ? vw : 0.f);
                }
            }
        };
        const int nstep = NRP / 16;
        const unsigned char* r0 = rimg + ql * rpitch + 16 * g, *r1 = r0 + 32 * rpitch;
        auto rfrag = [&](const unsigned char* rp, int s) { return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(rp + 32 * s)); };
        bf16x8 rf[2] = {rfrag(r0, 0), rfrag(r1, 0)};
        float gcur[8];
        gather(0, gcur);
        auto rstep = [&](int s, uint4& keep) {
            const int sn = min(s + 1, nstep - 1);
            const bf16x8 rn[2] = {rfrag(r0, sn), rfrag(r1, sn)};
            float gnext[8];
            if (abl & 32) {
#pragma unroll
                for (int t = 0; t < 8; ++t) gnext[t] = gcur[t] + 1.f;
            } else {
                gather(sn, gnext);
            }
            const bf16x8 gf = packfrag(gcur);
            if constexpr (FUSE) keep = __builtin_bit_cast(uint4, gf);
            else if (!(abl & 2)) *reinterpret_cast<uint4*>(dgrow + 16 * s + 8 * g) = __builtin_bit_cast(uint4, gf);
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                if (!(abl & 64)) dq[db] = mfma(rf[db], gf, dq[db]);
                rf[db] = rn[db];
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) gcur[t] = gnext[t];
        };
        if constexpr (FUSE) {           // fully unrolled: gfs[] has to stay in registers
#pragma unroll
            for (int s = 0; s < NSMAX; ++s)
                if (s < ((abl & 1) ? 0 : nstep)) rstep(s, gfs[s]);
        } else {
            uint4 none;
            for (int s = 0; s < ((abl & 1) ? 0 : nstep); ++s) rstep(s, none);
        }
        stage_rows(stg, dq, 1.f, lane);
        if (!(abl & 4)) write_rows(stg, dqkv + (size_t)(b * L + qt * 32) * ldq + h * ATT_HD, ldq, lane);
    }
    if constexpr (FUSE) {
        __syncthreads();                                     // every wave is done with the tables, Rcat^T and its dQ staging tile
        const int nimg = (NRP + 63) >> 6;                    // per wave: nimg images [32 q][64 r] of dG, then one [32 q][64 d] image of Q
        unsigned char* wimg = smem + wave * (nimg + 1) * IMG;
        if (valid) {
            const int rsw = vsw(ql);
#pragma unroll
            for (int s = 0; s < NSMAX; ++s)
                if (s < NRP / 16) *reinterpret_cast<uint4*>(wimg + (s >> 2) * IMG + ql * 128 + ((((2 * s + g) & 7) ^ rsw) << 4)) = gfs[s];
#pragma unroll
            for (int s = 0; s < 4; ++s)
                *reinterpret_cast<uint4*>(wimg + nimg * IMG + ql * 128 + (((2 * s + g) ^ rsw) << 4)) = __builtin_bit_cast(uint4, qf[s]);
        }
        __syncthreads();
        // 2 * NRP / 32 units (r-block, d-block) of one 32 x 32 accumulator over the workgroup's query blocks; wave w takes units w, w + 4, ...
        // Orientation: A = dG^T (rows r -> registers), B = Q^T (lanes = d), so that a store instruction writes 32 consecutive d of one row:
        // two 128-byte segments per instruction (the first form had lane = r: 64 rows, 16 bytes each, per instruction).
        float* pw = part + (size_t)(pslot0 + blockIdx.x) * NRP * ATT_HD;
        for (int u = wave; u < 2 * (NRP / 32); u += NW) {
            const int rb = u >> 1, db = u & 1;
            f32x16 acc = zero16();
            for (int w2 = 0; w2 < NW; ++w2) {
                if ((tile0 + blk * NW + w2) * 32 >= L) break;         // that wave had no queries
                const unsigned char* im = smem + w2 * (nimg + 1) * IMG;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    acc = mfma(trfrag(im + (rb >> 1) * IMG, la, rb & 1, ks), trfrag(im + nimg * IMG, la, db, ks), acc);
            }
            float* pcol = pw + (size_t)(rb * 32 + 4 * g) * ATT_HD + db * 32 + ql;          // D: registers = r, lane = d
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) pcol[((reg & 3) + 8 * (reg >> 2)) * ATT_HD] = acc[reg];
        }
    }
    if constexpr (TR) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    coarse(63);
}

// =============================================================================================== backward: dK, dV
// stage = [Q img | dO img | kw part of the T tile, chunk-swizzled [32 q][64 B] | 8 kh-table rows x [32 q] bf16 (6 key rows from the
// workgroup's first one, then -lse/scale hi, lo) | -Delta f32 [32]]
constexpr int DKV_TW = 2 * IMG, DKV_TH = DKV_TW + 2048, DKV_ND = DKV_TH + 512, DKV_STAGE = DKV_ND + 128;
template <int MINW>
__global__ __launch_bounds__(NT, MINW) void bwd_dkv_kernel(const bf16* __restrict__ qkv, size_t ldq, const bf16* __restrict__ dout,
                                                           size_t lddo, const unsigned char* __restrict__ tables, bf16* __restrict__ dqkv,
                                                           int L, int H, int Hp, float scale, int nblk, int xcd_map, int abl, int tile0) {
    // abl (diagnostics, PA_ATTN3_DKV_ABL; results WRONG when set): 16 no query loop
    // tile0: first 32-key tile of every head this launch covers (0 in the product)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 5, ql = lane & 31;
    int blk, bh;
    wg_coords(nblk, xcd_map, blk, bh, (L / 32 - tile0) / NW);
    const int b = bh / H, h = bh % H, D = H * ATT_HD;
    const bf16* qbase = qkv + (size_t)b * L * ldq + h * ATT_HD;
    const bf16* dobase = dout + (size_t)b * L * lddo + h * ATT_HD;
    const int kt = tile0 + blk * NW + wave;
    const bool valid = kt * 32 < L;
    const int key = kt * 32 + ql;
    const int kh = key / WP, kw = key % WP;
    const int khlo = ((tile0 + blk * NW) * 32) / WP;  // first key row of the workgroup
    const int kha = (kt * 32) / WP;                   // the wave's 32 keys lie in key rows kha and kha + 1
    const int ntile = L / 32;
    const int TB = ttile_bytes(Hp);
    const unsigned char* tbase = tables + (size_t)bh * ntile * TB;
    LaneAddr la;
    la.init(lane);
    EAddr ea;                                        // the kw part of the T tile uses the layout of the one-hot images
    ea.init(lane);
    const float sl = scale * LOG2E_F;
    const int sa = kha & 3, sb = (kha + 1) & 3;      // window slots of the wave's two key rows; the other two carry lse hi / lo

    // B operands that never change: K, V rows and the one-hot row of this lane's key
    bf16x8 kf[4], vf[4], eb0, eb1;
    int woff[2];
    {
        uint32_t e0[4] = {0, 0, 0, 0}, e1[4] = {0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const uint32_t bit = 0x3F80u << (16 * (t & 1));
            if (kw == 8 * g + t) e0[t >> 1] |= bit;
            if (t < 6) {
                if (kw == 16 + 6 * g + t) e1[t >> 1] |= bit;
            } else {
                const int slot = 2 * g + (t - 6);
                if (slot == (kh & 3) || (slot != sa && slot != sb)) e1[t >> 1] |= bit;
            }
        }
        eb0 = __builtin_bit_cast(bf16x8, make_uint4(e0[0], e0[1], e0[2], e0[3]));
        eb1 = __builtin_bit_cast(bf16x8, make_uint4(e1[0], e1[1], e1[2], e1[3]));
        // A-operand side (lane = query of the streamed tile): which staged kh-table row feeds this half-wave's two window slots
#pragma unroll
        for (int hs = 0; hs < 2; ++hs) {
            const int slot = 2 * g + hs;
            int row;
            if (slot == sa) row = kha - khlo;
            else if (slot == sb) row = kha + 1 - khlo;
            else {
                int rank = 0;
                for (int s2 = 0; s2 < slot; ++s2) rank += (s2 != sa && s2 != sb) ? 1 : 0;
                row = 6 + rank;
            }
            woff[hs] = DKV_TH + row * 64 + ql * 2;
        }
    }
    if (valid) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            kf[s] = gfrag(qbase + D + (size_t)key * ldq, s, g);
            vf[s] = gfrag(qbase + 2 * D + (size_t)key * ldq, s, g);
        }
    }
    // staging: Q tile, dO tile, and one more 16-byte chunk for the first 168 threads (kw part 128, table rows 32, -Delta 8)
    Stager qs, dos;
    uint4 rx = zero4();
    int xsrc = 0, xdst = 0;
    if (tid < 128) {
        const int qi = tid >> 2, c = tid & 3;
        xsrc = tid * 16;
        xdst = DKV_TW + qi * 64 + ((c ^ ((qi >> 2) & 3)) << 4);
    } else if (tid < 160) {
        const int k = tid - 128, rs = k >> 2, c = k & 3;
        const int srow = rs < 6 ? min(khlo + rs, Hp - 1) : Hp + (rs - 6);
        xsrc = 2048 + srow * 64 + c * 16;
        xdst = DKV_TH + rs * 64 + c * 16;
    } else if (tid < 168) {
        xsrc = 2048 + (Hp + 2) * 64 + (tid - 160) * 16;
        xdst = DKV_ND + (tid - 160) * 16;
    }
    auto load_all = [&](int j) {
        qs.load(qbase + (size_t)j * 32 * ldq, ldq, tid);
        dos.load(dobase + (size_t)j * 32 * lddo, lddo, tid);
        if (tid < 168) rx = *reinterpret_cast<const uint4*>(tbase + (size_t)j * TB + xsrc);
    };
    auto store_all = [&](int stage) {
        unsigned char* s0 = smem + stage * DKV_STAGE;
        qs.store(s0, tid);
        dos.store(s0 + IMG, tid);
        if (tid < 168) *reinterpret_cast<uint4*>(s0 + xdst) = rx;
    };
    load_all(0);
    store_all(0);
    __syncthreads();

    f32x16 dk[2], dv[2];
    dk[0] = zero16(); dk[1] = zero16(); dv[0] = zero16(); dv[1] = zero16();

    for (int j = 0; j < ((abl & 16) ? 0 : ntile); ++j) {
        if (j + 1 < ntile) load_all(j + 1);
        const unsigned char* qimg = smem + (j & 1) * DKV_STAGE;
        const unsigned char* doimg = qimg + IMG;
        if (valid) {
            const uint4 a0 = *reinterpret_cast<const uint4*>(qimg + DKV_TW + ea.row[0]);
            uint4 a1 = *reinterpret_cast<const uint4*>(qimg + DKV_TW + ea.row[1]);
            const uint32_t wlo = *reinterpret_cast<const uint16_t*>(qimg + woff[0]);
            const uint32_t whi = *reinterpret_cast<const uint16_t*>(qimg + woff[1]);
            f32x16 dpacc;
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const float4 nd = *reinterpret_cast<const float4*>(qimg + DKV_ND + (8 * rg + 4 * g) * 4);
                dpacc[rg * 4 + 0] = nd.x; dpacc[rg * 4 + 1] = nd.y; dpacc[rg * 4 + 2] = nd.z; dpacc[rg * 4 + 3] = nd.w;
            }
            bf16x8 qfr[4], dofr[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) { qfr[s] = rowfrag(qimg, la, s); dofr[s] = rowfrag(doimg, la, s); }
            if constexpr (MINW < 3 && A3_HOIST) __builtin_amdgcn_sched_barrier(0);
            a1.w = wlo | (whi << 16);
            f32x16 sacc = zero16();
            A3_PRIO_UP();
            sacc = mfma(as_frag(a0), eb0, sacc);                          // S[q][key] - lse: lane = key, registers = q rows
            sacc = mfma(as_frag(a1), eb1, sacc);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                sacc = mfma(qfr[s], kf[s], sacc);
                dpacc = mfma(dofr[s], vf[s], dpacc);                      // dP[q][key] - Delta[q]
            }
            A3_PRIO_DOWN();
            bf16x8 dotr[2][2], qtr[2][2];
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                dotr[db][0] = trfrag(doimg, la, db, 0); dotr[db][1] = trfrag(doimg, la, db, 1);
                qtr[db][0] = trfrag(qimg, la, db, 0); qtr[db][1] = trfrag(qimg, la, db, 1);
            }
            if constexpr (MINW < 3 && A3_HOIST) __builtin_amdgcn_sched_barrier(0);
            float p[16], ds[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                p[r] = __builtin_amdgcn_exp2f(sacc[r] * sl);
                ds[r] = p[r] * dpacc[r];
            }
            const bf16x8 pf0 = packfrag(p), pf1 = packfrag(p + 8), dsf0 = packfrag(ds), dsf1 = packfrag(ds + 8);
            A3_PRIO_UP();
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                dv[db] = mfma(dotr[db][0], pf0, dv[db]);    // dV^T[d][key] += dO^T[d][q] P[q][key]
                dv[db] = mfma(dotr[db][1], pf1, dv[db]);
                dk[db] = mfma(qtr[db][0], dsf0, dk[db]);    // dK^T[d][key] += Q^T[d][q] dS[q][key]
                dk[db] = mfma(qtr[db][1], dsf1, dk[db]);
            }
            A3_PRIO_DOWN();
        }
        if (j + 1 < ntile) store_all((j + 1) & 1);
        __syncthreads();
    }
    unsigned char* stg = smem + wave * 2 * IMG;
    if (valid) {
        stage_rows(stg, dk, scale, lane);
        stage_rows(stg + IMG, dv, 1.f, lane);
        bf16* orow = dqkv + (size_t)(b * L + kt * 32) * ldq + h * ATT_HD;
        write_rows(stg, orow + D, ldq, lane);
        write_rows(stg + IMG, orow + 2 * D, ldq, lane);
    }
}

// -lse / scale as a bf16 hi + lo pair and -Delta into the table tiles (one thread per (sample, head, query))
__global__ void prep_kernel(const float* __restrict__ lse, const float* __restrict__ delta, unsigned char* __restrict__ tables, int L, int Hp,
                            float inv_scale, int total) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int bh = idx / L, q = idx - bh * L, qt = q >> 5, qi = q & 31;
    unsigned char* tt = tables + ((size_t)bh * (L / 32) + qt) * ttile_bytes(Hp);
    const float x = -lse[idx] * inv_scale;
    const bf16 hi = (bf16)x;
    const bf16 lo = (bf16)(x - (float)hi);
    *reinterpret_cast<bf16*>(tt + 2048 + Hp * 64 + qi * 2) = hi;
    *reinterpret_cast<bf16*>(tt + 2048 + (Hp + 1) * 64 + qi * 2) = lo;
    *reinterpret_cast<float*>(tt + 2048 + (Hp + 2) * 64 + qi * 4) = -delta[idx];
}

// Delta = rowsum(dO o O) and the prep above in ONE pass (the backward of a block used to launch attn_delta_kernel, then prep_kernel):
// one wave per token row, 4 lanes per head (16 elements each), the quad's first lane writes the head's three table fields
__global__ __launch_bounds__(256) void prep_delta_kernel(const bf16* __restrict__ o, size_t ldo, const bf16* __restrict__ d_o, size_t lddo,
                                                         const float* __restrict__ lse, unsigned char* __restrict__ tables, int R, int L, int H,
                                                         int Hp, float inv_scale) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= R) return;
    const int b = row / L, l = row - b * L, qt = l >> 5, qi = l & 31;
    const int D = H * ATT_HD, TB = ttile_bytes(Hp);
    for (int c = lane * 16; c < D; c += 1024) {
        const uint4 o0 = *reinterpret_cast<const uint4*>(o + (size_t)row * ldo + c), o1 = *reinterpret_cast<const uint4*>(o + (size_t)row * ldo + c + 8);
        const uint4 d0 = *reinterpret_cast<const uint4*>(d_o + (size_t)row * lddo + c), d1 = *reinterpret_cast<const uint4*>(d_o + (size_t)row * lddo + c + 8);
        float s = 0.f;          // the element order of attn_delta_kernel (the same bits as the two-kernel route)
        const uint32_t ow[8] = {o0.x, o0.y, o0.z, o0.w, o1.x, o1.y, o1.z, o1.w}, dw[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            s += bf16_lo(ow[e]) * bf16_lo(dw[e]);
            s += bf16_hi(ow[e]) * bf16_hi(dw[e]);
        }
        s = quad_sum(s);
        if ((lane & 3) == 0) {
            const int bh = b * H + c / ATT_HD;
            unsigned char* tt = tables + ((size_t)bh * (L / 32) + qt) * TB;
            const float x = -lse[(size_t)bh * L + l] * inv_scale;
            const bf16 hi = (bf16)x;
            const bf16 lo = (bf16)(x - (float)hi);
            *reinterpret_cast<bf16*>(tt + 2048 + Hp * 64 + qi * 2) = hi;
            *reinterpret_cast<bf16*>(tt + 2048 + (Hp + 1) * 64 + qi * 2) = lo;
            *reinterpret_cast<float*>(tt + 2048 + (Hp + 2) * 64 + qi * 4) = -s;
        }
    }
}

}   // namespace a3

int attn3_bwd_prep(const bf16* out, int64_t ldo, const bf16* dout, int64_t lddo, const float* lse, void* tables, int Bn, int L, int H, int Hp,
                   float scale, hipStream_t st) {
    const int R = Bn * L;
    PA_LAUNCH(a3::prep_delta_kernel, dim3((R + 3) / 4), dim3(256), 0, st, out, (size_t)ldo, dout, (size_t)lddo, lse,
              reinterpret_cast<unsigned char*>(tables), R, L, H, Hp, 1.f / scale);
    return (int)hipGetLastError();
}

// PA_ATTN3=0 / pa_attn_set_generation(2): keep the generation-2 kernels for every grid (A/B runs, cross-generation tests).  The paired
// 8-wave build and the software-pipelined dQ of round 2 both measured slower and live in tools/experiments/ (DESIGN.md section 4.5).
static int g_attn_generation = 0;
extern "C" int pa_attn_set_generation(int generation) {
    if (generation != 0 && generation != 2 && generation != 3) return (int)hipErrorInvalidValue;
    g_attn_generation = generation;
    return 0;
}
static int g_attn_trace = 0;
extern "C" int pa_attn_trace(int enable, unsigned long long* host_out) {
    g_attn_trace = enable;
    if (host_out != nullptr) return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(a3::g_trace), sizeof(a3::g_trace));
    return 0;
}
bool attn3_ok(int L, int Hp, int Wp) {
    static const int on = [] { const char* e = getenv("PA_ATTN3"); return e ? atoi(e) : 1; }();
    return on && g_attn_generation != 2 && Wp == a3::WP && Hp % a3::RPP == 0 && Hp >= a3::RPP && L == Hp * Wp;
}
int64_t attn3_table_bytes(int Bn, int L, int H, int Hp, int Wp) {
    if (!attn3_ok(L, Hp, Wp)) return 0;
    return (int64_t)Bn * H * (L / 32) * a3::ttile_bytes(Hp);
}
// bit 0: XCD-contiguous head order (PA_ATTN_XCD=0 turns it off); bit 1: light workgroups (the 13th of every head: one live wave of four)
// dispatched last -- OFF by default since round 5 (PA_ATTN_LIGHT_LAST=1 / pa_debug_set(8, 2) turn it on): interleaved A/B on the whole
// training step 53.33 vs 53.41 ms (profiles/r05_ab_light_workgroups_last.log: inside the noise), while the light workgroups, run last,
// find their head's K / V evicted from the XCD's L2: +50 MB per launch of fabric reads, forward and backward.
// A change that adds bytes and buys nothing does not stay.
static int a3_xcd_map_on() {
    static const int v = [] {
        const char* e = getenv("PA_ATTN_XCD");
        const char* l = getenv("PA_ATTN_LIGHT_LAST");
        return ((e ? atoi(e) : 1) ? 1 : 0) | ((l ? atoi(l) : 0) ? 2 : 0);
    }();
    if (g_attn_light_last == 1) return v & 1;
    if (g_attn_light_last == 2) return v | 2;
    return v;
}

int attn3_fwd(const bf16* qkv, int64_t ldq, const bf16* rcat, bf16* out, int64_t ldo, float* lse, void* tables, int Bn, int L, int H,
              int Hp, int Wp, float scale, hipStream_t st) {
    using namespace a3;
    const int NRP = pa_relpos_rows_padded(Hp, Wp);
    // PA_ATTN3_FWD_STAGES: 1 (default) = single K/V stage, 4 workgroups per CU; 2 = double-buffered, one barrier per tile
    static const int stages = [] { const char* v = getenv("PA_ATTN3_FWD_STAGES"); return v ? atoi(v) : 1; }();
    const size_t smem = (size_t)(stages == 2 ? 2 : 1) * STAGE_QK + (size_t)NW * Hp * 64 + PH * EIMG;
    static bool done1 = false, done2 = false;
    auto kern = stages == 2 ? fwd_kernel<2> : fwd_kernel<1>;
    if (int e = set_smem(reinterpret_cast<const void*>(kern), stages == 2 ? done2 : done1)) return e;
    const int nblk = (L / 32 + NW - 1) / NW;
    PA_LAUNCH(kern, dim3(nblk * Bn * H), dim3(NT), smem, st, qkv, (size_t)ldq, rcat, out, (size_t)ldo, lse,
              reinterpret_cast<unsigned char*>(tables), L, H, Hp, NRP, scale, nblk, a3_xcd_map_on(), [] { const char* v = getenv("PA_ATTN3_FWD_ABL"); return v ? atoi(v) : 0; }());
    return (int)hipGetLastError();
}

// ---- fused rel-pos table gradient: one fp32 [NRP][64] partial per dQ workgroup, summed in a fixed order (two stages)
namespace a3 {
constexpr int RED_ZC = 32;
__global__ __launch_bounds__(256) void relpos_part_reduce_kernel(const float4* __restrict__ part, float4* __restrict__ tmp, int n4, int nz) {
    __shared__ float4 sh[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), zl = threadIdx.x >> 6;
    const int z0 = (int)((int64_t)nz * blockIdx.y / gridDim.y), z1 = (int)((int64_t)nz * (blockIdx.y + 1) / gridDim.y);
    float4 s = make_float4(0, 0, 0, 0);
    if (c < n4) {
#pragma unroll 4
        for (int z = z0 + zl; z < z1; z += 4) {
            const float4 v = part[(size_t)z * n4 + c];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    sh[zl][threadIdx.x & 63] = s;
    __syncthreads();
    if (zl == 0 && c < n4) {
        float4 t = sh[0][threadIdx.x];
#pragma unroll
        for (int k = 1; k < 4; ++k) { const float4 v = sh[k][threadIdx.x]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        tmp[(size_t)blockIdx.y * n4 + c] = t;
    }
}
}   // namespace a3
extern "C" int pa_slab_reduce(const float* in, float* out, int64_t n, int nz, int64_t stride, int accumulate, hipStream_t st);
static int a3_fuse_on() {
    static const int v = [] { const char* e = getenv("PA_ATTN3_FUSE_RELPOS"); return e ? atoi(e) : 1; }();
    return g_attn3_fuse == 1 ? 0 : (g_attn3_fuse == 2 ? 1 : v);
}
// workgroups of the dQ launch = partial slots
static int a3_blocks(int L) { return (L / 32 + a3::NW - 1) / a3::NW; }
static int64_t a3_num_partials(int Bn, int L, int H, int, int) { return (int64_t)a3_blocks(L) * Bn * H; }
int64_t attn3_relpos_partials_bytes(int Bn, int L, int H, int Hp, int Wp) {
    if (!a3_fuse_on() || !attn3_ok(L, Hp, Wp)) return 0;
    const int NRP = pa_relpos_rows_padded(Hp, Wp);
    if (NRP > 16 * a3::NSMAX) return 0;
    return a3_num_partials(Bn, L, H, Hp, Wp) * NRP * ATT_HD * sizeof(float);
}
// part: what attn3_bwd's dQ kernel wrote; tmp: RED_ZC * NRP * 64 floats of scratch; drcat f32 [NRP][64], overwritten
int attn3_relpos_reduce(const float* part, float* drcat, float* tmp, int Bn, int L, int H, int Hp, int Wp, hipStream_t st) {
    using namespace a3;
    const int NRP = pa_relpos_rows_padded(Hp, Wp);
    const int nz = (int)a3_num_partials(Bn, L, H, Hp, Wp), n4 = NRP * ATT_HD / 4;
    const int zc = nz < RED_ZC ? nz : RED_ZC;
    PA_LAUNCH(relpos_part_reduce_kernel, dim3((n4 + 63) / 64, zc), dim3(256), 0, st, reinterpret_cast<const float4*>(part),
              reinterpret_cast<float4*>(tmp), n4, nz);
    if (int e = (int)hipGetLastError()) return e;
    return pa_slab_reduce(tmp, drcat, (int64_t)n4 * 4, zc, (int64_t)n4 * 4, 0, st);
}

int attn3_bwd(const bf16* qkv, int64_t ldq, const bf16* rcatT, const bf16* dout, int64_t lddo, const float* lse, const float* delta,
              void* tables, bf16* dqkv, bf16* dG, float* part, int Bn, int L, int H, int Hp, int Wp, float scale, const bf16* out, int64_t ldo,
              hipStream_t st) {
    using namespace a3;
    const int NRP = pa_relpos_rows_padded(Hp, Wp);
    if (part != nullptr && NRP > 16 * NSMAX) return (int)hipErrorInvalidValue;
    if (part == nullptr && dG == nullptr) return (int)hipErrorInvalidValue;
    unsigned char* tb = reinterpret_cast<unsigned char*>(tables);
    int e;
    // delta given: the two-kernel route (pa_attn_bwd_delta + the prep kernel here).  delta NULL: with `out` the dQ kernel computes Delta itself
    // and the forward has written the lse fields (no extra launch: the default since round 5); without `out`, pa_attn_bwd_prep has filled both
    if (delta != nullptr) {
        const int total = Bn * H * L;
        PA_LAUNCH(prep_kernel, dim3((total + 255) / 256), dim3(256), 0, st, lse, delta, tb, L, Hp, 1.f / scale, total);
        if ((e = (int)hipGetLastError())) return e;
    }
    // (the 64-row-wave generation 4 of round 4, which took whole 8-tile groups of every head in front of these launches, measured 40 %
    // slower and lives on the branch exp/attn4-generation-4: DESIGN.md section 4.2d)
    const int tile0 = 0;
    const int nblk = a3_blocks(L);
    // PA_ATTN3_DQ_WAVES / PA_ATTN3_DKV_WAVES: waves per SIMD the register allocation aims at (2 or 3)
    static const int dq_w = [] { const char* v = getenv("PA_ATTN3_DQ_WAVES"); return v ? atoi(v) : 2; }();
    static const int dkv_w = [] { const char* v = getenv("PA_ATTN3_DKV_WAVES"); return v ? atoi(v) : 2; }();
    if (nblk > 0) {
        static const size_t pad = [] { const char* v = getenv("PA_ATTN3_LDS_PAD"); return v ? (size_t)atoi(v) : (size_t)0; }();   // diagnostics: fewer workgroups per CU
        size_t smem = 2 * (size_t)STAGE_QK + (size_t)NW * Hp * 64 + PH * EIMG + (size_t)ATT_HD * (NRP * 2 + 16) + pad;
        const bool fuse = part != nullptr;
        if (fuse) {                                             // the dG / Q images of the fused rel-pos contraction reuse the whole allocation
            const size_t need = (size_t)NW * ((NRP + 63) / 64 + 1) * IMG;
            if (smem < need) smem = need;
        }
        auto kern = fuse ? bwd_dq_kernel<2, A3_NDL, false, true>
                         : (g_attn_trace ? bwd_dq_kernel<2, true, true> : (dq_w == 3 ? bwd_dq_kernel<3, false> : bwd_dq_kernel<2, A3_NDL>));
        static bool done2 = false, done3 = false, donet = false, donef = false;
        if ((e = set_smem(reinterpret_cast<const void*>(kern), fuse ? donef : (g_attn_trace ? donet : (dq_w == 3 ? done3 : done2))))) return e;
        const char* ablv = getenv("PA_ATTN3_DQ_ABL");           // diagnostics, read per launch
        PA_LAUNCH(kern, dim3(nblk * Bn * H), dim3(NT), smem, st, qkv, (size_t)ldq, rcatT, dout, (size_t)lddo, lse, tb, dqkv, dG, part, L, H,
                  Hp, NRP, scale, nblk, a3_xcd_map_on(), ablv ? atoi(ablv) : 0, tile0, 0, delta == nullptr ? out : nullptr, (size_t)ldo);
        if ((e = (int)hipGetLastError())) return e;
    }
    const int tile0_kv = 0, nblk_kv = a3_blocks(L);
    if (nblk_kv > 0) {
        size_t smem = 2 * (size_t)DKV_STAGE;
        if (smem < (size_t)NW * 2 * IMG) smem = (size_t)NW * 2 * IMG;
        static const size_t pad = [] { const char* v = getenv("PA_ATTN3_LDS_PAD"); return v ? (size_t)atoi(v) : (size_t)0; }();
        smem += pad;
        auto kern = dkv_w == 3 ? bwd_dkv_kernel<3> : bwd_dkv_kernel<2>;
        static bool done2 = false, done3 = false;
        if ((e = set_smem(reinterpret_cast<const void*>(kern), dkv_w == 3 ? done3 : done2))) return e;
        PA_LAUNCH(kern, dim3(nblk_kv * Bn * H), dim3(NT), smem, st, qkv, (size_t)ldq, dout, (size_t)lddo, tb, dqkv, L, H, Hp, scale, nblk_kv,
                  a3_xcd_map_on(), [] { const char* v = getenv("PA_ATTN3_DKV_ABL"); return v ? atoi(v) : 0; }(), tile0_kv);
        if ((e = (int)hipGetLastError())) return e;
    }
    return 0;
}
