// Generation-3 attention, paired build: 8-wave workgroups whose two half-groups alternate a MATRIX phase and a VALU phase in anti-phase.
// Same math, tables and tile images as attn3.hip (Painter/models_painter.py:76-86, util/vitdet_utils.py:63-125).
//
// Why.  SQ counters of the 4-wave build (profiles/r02_attn_sq_counters_gen2_vs_gen3_4wave.json): per 32x32 tile a wave spends ~520 cycles
// on the matrix pipe and ~570 on the VALU (16 v_exp at a quarter rate are half of that), yet a tile pair costs ~1270 cycles per SIMD --
// the two resident waves of a SIMD belong to different workgroups, drift, and ask for the same pipe at the same time (MFMA 31 %, VALU 35 %
// busy).  Here the two waves of a SIMD are waves w and w + 4 of ONE workgroup and are kept half a tile apart by barriers:
//
//     phase 2t     group 0: M(t)    matrix work: second MFMA group of tile t-1, first MFMA group of tile t, all LDS fragment reads
//                  group 1: V(t-1)  VALU work of tile t-1 (exp, products, packing), staging traffic, table write-backs
//     phase 2t+1   group 0: V(t)                 group 1: M(t)
//
// M and V cost about the same, so each SIMD has its matrix pipe and its VALU busy at once, by construction.  The split of a tile's
// MFMAs is by dependence: the first group (logits, dP) feeds V, the second (P.V, or dQ / dK / dV and the bias gradient) consumes V's
// packed operands one phase later, so no accumulator is duplicated.  K/V (or Q/dO/table) tiles live in a ring of three LDS stages:
// M(t) reads tiles t-1 and t while tile t+1 (t+2 for group 1) is being written; group 0's threads stage the first image of a tile
// (K, or Q + the kw part of the table tile), group 1's the second (V, or dO + the kh rows and -Delta), register-staged one tile ahead.
// Barriers are raw s_barrier with an explicit lgkmcnt(0) -- a __syncthreads() would also drain the staging loads in flight.
// Workgroup = 8 waves x 32 rows = 256 rows of the stationary axis; 49 row tiles = 7 workgroups per head, the last wave of each idle.
#include "attn3_common.h"
#include "../../include/painter_hip.h"
#include "attn3.h"
#include <cstdlib>

namespace a3 {
namespace pr {

constexpr int NWP = 8, NTP = 512, RING = 3;

DEVI void phase_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's LDS writes have landed, its reads have returned
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}
DEVI int ring_next(int s) { return s == RING - 1 ? 0 : s + 1; }

#define A3P_FOR_PHASES(body, a)                                                                                          \
    body(std::integral_constant<int, 0>{}, a); body(std::integral_constant<int, 1>{}, a); body(std::integral_constant<int, 2>{}, a); \
    body(std::integral_constant<int, 3>{}, a); body(std::integral_constant<int, 4>{}, a); body(std::integral_constant<int, 5>{}, a); \
    body(std::integral_constant<int, 6>{}, a);

// =============================================================================================== forward
// LDS: thT 8 waves x [Hp][32 q] bf16 | ring 3 x [K img | V img] | 7 one-hot images  (the per-wave T rows alias the ring in the prologue)
__global__ __launch_bounds__(NTP, 2) void fwd_kernel(const bf16* __restrict__ qkv, size_t ldq, const bf16* __restrict__ rcat, bf16* __restrict__ out,
                                                     size_t ldo, float* __restrict__ lse, unsigned char* __restrict__ tables, int L, int H, int Hp,
                                                     int NRP, float scale, int nblk, int xcd_map) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 5, ql = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), grp = wave >> 2, st = tid & 255;
    int blk, bh;
    wg_coords(nblk, xcd_map, blk, bh);
    const int b = bh / H, h = bh % H, D = H * ATT_HD;
    const bf16* base = qkv + (size_t)b * L * ldq + h * ATT_HD;
    const bf16* sbase = base + (grp ? 2 * D : D);           // what this thread stages: group 0 K tiles, group 1 V tiles
    const int soff = grp ? IMG : 0;
    const int qt = blk * NWP + wave;
    const bool valid = qt * 32 < L;
    const int q = qt * 32 + ql;
    unsigned char* thT = smem + wave * Hp * 64;
    unsigned char* ring = smem + NWP * Hp * 64;
    unsigned char* eimg = ring + RING * STAGE_QK;
    unsigned char* twimg = ring + wave * 2048;
    LaneAddr la;
    la.init(lane);
    EAddr ea;
    ea.init(lane);
    const int ntile = L / 32;
    Stager sg;
    sg.load(sbase, ldq, st);

    bf16x8 qf[4];
    uint4 T0 = zero4(), T1 = zero4();
    if (valid) {
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = gfrag(base + (size_t)q * ldq, s, g);
        build_tables3(twimg, thT, rcat, NRP, qf, q / WP, q % WP, Hp, 1.f / scale, lane, eimg + tid * 2);      // trash: the image region, built later
        T0 = *reinterpret_cast<const uint4*>(twimg + ql * 64 + 16 * g);          // same-wave LDS ops are ordered
        T1 = *reinterpret_cast<const uint4*>(twimg + ql * 64 + 32 + 16 * g);
        if (tables != nullptr) {
            unsigned char* tt = tables + ((size_t)bh * ntile + qt) * ttile_bytes(Hp);
#pragma unroll
            for (int i = 0; i < 2; ++i) *reinterpret_cast<uint4*>(tt + (lane + 64 * i) * 16) = *reinterpret_cast<const uint4*>(twimg + (lane + 64 * i) * 16);
            for (int c = lane; c < Hp * 4; c += 64) *reinterpret_cast<uint4*>(tt + 2048 + c * 16) = *reinterpret_cast<const uint4*>(thT + c * 16);
        }
    }
    __syncthreads();                    // the T rows in the ring are dead
    build_eimg(eimg, tid, NTP);
    sg.store(ring + soff, st);                                              // tile 0
    sg.load(sbase + (size_t)min(1, ntile - 1) * 32 * ldq, ldq, st);         // tile 1: group 0 keeps it in registers until V(0)
    if (grp) {
        sg.store(ring + STAGE_QK + soff, st);                               // group 1 writes its half of tile 1 now and holds tile 2
        sg.load(sbase + (size_t)min(2, ntile - 1) * 32 * ldq, ldq, st);
    }
    __syncthreads();
    if (grp) phase_barrier();           // group 1 runs one phase behind

    f32x16 oacc[2], sacc = zero16();
    oacc[0] = zero16();
    oacc[1] = zero16();
    bf16x8 pf0, pf1;
    float m = 0.f, l = 0.f;
    const float sl = scale * LOG2E_F;
    const unsigned char* thw = thT + ql * 2;
    int slot = 0;                       // ring stage of the current tile

    auto body = [&](auto pc, int a) {
        constexpr int P = decltype(pc)::value;
        const int t = a * PH + P;
        const int sprev = slot == 0 ? RING - 1 : slot - 1;
        // ---------------------------------------------------------------- M(t): P.V of tile t-1, logits of tile t
        if (valid) {
            const unsigned char* kimg = ring + slot * STAGE_QK;
            const unsigned char* vimg = ring + sprev * STAGE_QK + IMG;
            const unsigned char* thr = thw + a * (RPP * 64);
            if constexpr (P == 0) win_set<0>(T1.w, thr);
            win_set<(P + 1) & 1>(T1.w, thr + (P + 1) * 64);
            const unsigned char* ei = eimg + P * EIMG;
            bf16x8 vtr[2][2], kfr[4];
            if (t > 0) {
#pragma unroll
                for (int db = 0; db < 2; ++db) { vtr[db][0] = trfrag(vimg, la, db, 0); vtr[db][1] = trfrag(vimg, la, db, 1); }
            }
            const bf16x8 ef0 = efrag(ei, ea, 0), ef1 = efrag(ei, ea, 1);
#pragma unroll
            for (int s = 0; s < 4; ++s) kfr[s] = rowfrag(kimg, la, s);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
            if (t > 0) {
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    oacc[db] = mfma(vtr[db][0], pf0, oacc[db]);
                    oacc[db] = mfma(vtr[db][1], pf1, oacc[db]);
                }
            }
            sacc = mfma(ef0, as_frag(T0), zero16());
            sacc = mfma(ef1, as_frag(T1), sacc);
#pragma unroll
            for (int s = 0; s < 4; ++s) sacc = mfma(kfr[s], qf[s], sacc);
            __builtin_amdgcn_s_setprio(0);
        }
        phase_barrier();
        // ---------------------------------------------------------------- V(t): staging traffic, then the softmax of tile t
        {
            const int tw = t + 1 + grp;                     // the tile this thread's registers hold
            if (tw < ntile) {
                int sw = ring_next(slot);
                if (grp) sw = ring_next(sw);
                sg.store(ring + sw * STAGE_QK + soff, st);
            }
            sg.load(sbase + (size_t)min(tw + 1, ntile - 1) * 32 * ldq, ldq, st);
        }
        if (valid) {
            float p[16];
            const float nm = -m;
#pragma unroll
            for (int r = 0; r < 16; ++r) p[r] = fmaf(sacc[r], sl, nm);
            float tmax = max16(p);
            tmax = fmaxf(tmax, xor32(tmax));
            if (t == 0 || __any(tmax > THR)) {          // wave-uniform; after the first tiles almost never taken
                const float delta = (t == 0) ? tmax : fmaxf(tmax, 0.f);
                const float alpha = __builtin_amdgcn_exp2f(-delta);
                m += delta;
                l *= alpha;
#pragma unroll
                for (int r = 0; r < 16; ++r) { oacc[0][r] *= alpha; oacc[1][r] *= alpha; p[r] -= delta; }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) p[r] = __builtin_amdgcn_exp2f(p[r]);
            l += sum16(p);
            pf0 = packfrag(p);
            pf1 = packfrag(p + 8);
        }
        slot = ring_next(slot);
        phase_barrier();
    };
    for (int a = 0; a < Hp / RPP; ++a) { A3P_FOR_PHASES(body, a) }
    // tail: P.V of the last tile
    if (valid) {
        const unsigned char* vimg = ring + (slot == 0 ? RING - 1 : slot - 1) * STAGE_QK + IMG;
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            oacc[db] = mfma(trfrag(vimg, la, db, 0), pf0, oacc[db]);
            oacc[db] = mfma(trfrag(vimg, la, db, 1), pf1, oacc[db]);
        }
    }
    if (!grp) phase_barrier();          // barrier counts of the two groups match again
    __syncthreads();                    // nobody reads the ring any more: per-wave 4 KB staging tiles (3 stages + part of the images)
    unsigned char* stg = ring + wave * IMG;
    if (valid) {
        const float lt = l + xor32(l);
        if (g == 0) lse[(size_t)bh * L + q] = (m + __builtin_amdgcn_logf(lt)) * LN2_F;
        stage_rows(stg, oacc, 1.f / lt, lane);
        write_rows(stg, out + (size_t)(b * L + qt * 32) * ldo + h * ATT_HD, ldo, lane);   // same-wave LDS ops are ordered
    }
}

// =============================================================================================== backward: dQ, bias gradients
// LDS: thT 8 waves x [Hp][32 q] bf16 (values, replaced row by row by their gradients) | ring 3 x [K img | V img] | 7 one-hot images
// diagnostics (PA_ATTN3_ABL bit 512): s_memtime stamps of workgroup 0, waves 0 (group 0) and 4 (group 1): [wave][0] kernel start, [1] loop
// start, [2] loop end, [3] kernel end, [8 + 4 t + k], t < 12: tile t's phase marks k = 0 top of M, 1 after M's MFMAs were issued, 2 after the
// barrier that ends M, 3 after V's work (before its barrier)
__device__ unsigned long long g_trace_p[2][64];
__global__ __launch_bounds__(NTP, 2) void bwd_dq_kernel(const bf16* __restrict__ qkv, size_t ldq, const bf16* __restrict__ rcatT,
                                                        const bf16* __restrict__ dout, size_t lddo, const float* __restrict__ lse,
                                                        const unsigned char* __restrict__ tables, bf16* __restrict__ dqkv, bf16* __restrict__ dG,
                                                        int L, int H, int Hp, int NRP, float scale, int nblk, int xcd_map, int abl) {
    // abl (diagnostics, PA_ATTN3_ABL; results are WRONG with any bit set): 1 no staging traffic, 4 no LDS fragment requests, 8 no exp / dS VALU,
    // 16 no table write-backs, 32 no MFMAs, 128 no phase barriers
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 5, ql = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), grp = wave >> 2, st = tid & 255;
    const bool trace = (abl & 512) && blockIdx.x == 0 && (wave & 3) == 0;
    auto stamp = [&](int slot) {
        if (trace) {
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (lane == 0 && slot < 64) g_trace_p[grp][slot] = t;
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    stamp(0);
    int blk, bh;
    wg_coords(nblk, xcd_map, blk, bh);
    const int b = bh / H, h = bh % H, D = H * ATT_HD;
    const bf16* base = qkv + (size_t)b * L * ldq + h * ATT_HD;
    const bf16* sbase = base + (grp ? 2 * D : D);
    const int soff = grp ? IMG : 0;
    const int qt = blk * NWP + wave;
    const bool valid = qt * 32 < L;
    const int q = qt * 32 + ql;
    const int qh = q / WP, qw = q % WP;
    unsigned char* thT = smem + wave * Hp * 64;
    unsigned char* ring = smem + NWP * Hp * 64;
    unsigned char* eimg = ring + RING * STAGE_QK;
    LaneAddr la;
    la.init(lane);
    EAddr ea;
    ea.init(lane);
    const int ntile = L / 32;
    const float sl = scale * LOG2E_F;
    Stager sg;
    sg.load(sbase, ldq, st);
    build_eimg(eimg, tid, NTP);

    bf16x8 qf[4], dof[4];
    uint4 T0 = zero4(), T1 = zero4();
    float nlse2 = 0.f, ndlt = 0.f;
    if (valid) {
        const unsigned char* tt = tables + ((size_t)bh * ntile + qt) * ttile_bytes(Hp);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            qf[s] = gfrag(base + (size_t)q * ldq, s, g);
            dof[s] = gfrag(dout + (size_t)(b * L + q) * lddo + h * ATT_HD, s, g);
        }
        T0 = *reinterpret_cast<const uint4*>(tt + ql * 64 + 16 * g);
        T1 = *reinterpret_cast<const uint4*>(tt + ql * 64 + 32 + 16 * g);
        for (int c = lane; c < Hp * 4; c += 64) *reinterpret_cast<uint4*>(thT + c * 16) = *reinterpret_cast<const uint4*>(tt + 2048 + c * 16);
        nlse2 = -lse[(size_t)bh * L + q] * LOG2E_F;
        ndlt = *reinterpret_cast<const float*>(tt + 2048 + (Hp + 2) * 64 + ql * 4);
    }
    f32x16 ndl;                         // -Delta in every register: C operand of the dP chain
#pragma unroll
    for (int r = 0; r < 16; ++r) ndl[r] = ndlt;
    sg.store(ring + soff, st);                                              // tile 0
    sg.load(sbase + (size_t)min(1, ntile - 1) * 32 * ldq, ldq, st);
    if (grp) {
        sg.store(ring + STAGE_QK + soff, st);
        sg.load(sbase + (size_t)min(2, ntile - 1) * 32 * ldq, ldq, st);
    }
    __syncthreads();
    if (grp) phase_barrier();           // group 1 runs one phase behind

    f32x16 dq[2], eacc, sacc = zero16(), dpacc = zero16();
    dq[0] = zero16();
    dq[1] = zero16();
    eacc = zero16();
    bf16x8 dsf0, dsf1;
    unsigned char* thw = thT + ql * 2;
    int slot = 0;

    // a completed key row's gradient (a D row of half-wave 1, see attn3.hip) replaces the table entry it came from
    auto write_back = [&](int slot4, int row) {
        if (g) {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
                if (s4 == slot4) {
                    *reinterpret_cast<bf16*>(thw + row * 64) = (bf16)eacc[win_reg(s4)];
                    eacc[win_reg(s4)] = 0.f;
                }
        }
    };

    stamp(1);
    auto body = [&](auto pc, int a) {
        constexpr int P = decltype(pc)::value;
        constexpr int PP = (P + PH - 1) % PH;               // phase of tile t-1
        const int t = a * PH + P;
        if (t < 12) stamp(8 + 4 * t);
        const int sprev = slot == 0 ? RING - 1 : slot - 1;
        // ---------------------------------------------------------------- M(t): dQ / bias-gradient MFMAs of tile t-1, S and dP of tile t
        if (valid) {
            const unsigned char* kimg = ring + slot * STAGE_QK;
            const unsigned char* vimg = kimg + IMG;
            const unsigned char* kprev = ring + sprev * STAGE_QK;
            const unsigned char* thr = thw + a * (RPP * 64);
            if constexpr (P == 0) win_set<0>(T1.w, thr);
            win_set<(P + 1) & 1>(T1.w, thr + (P + 1) * 64);
            const unsigned char* ei = eimg + P * EIMG;
            const unsigned char* eprev = eimg + PP * EIMG;
            bf16x8 ktr[2][2], etr[2], kfr[4], vfr[4];
            bf16x8 ef0, ef1;
            if (abl & 4) {
#pragma unroll
                for (int s = 0; s < 4; ++s) { kfr[s] = qf[s]; vfr[s] = dof[s]; }
                ktr[0][0] = ktr[0][1] = ktr[1][0] = ktr[1][1] = etr[0] = etr[1] = ef0 = ef1 = qf[0];
            } else {
            if (t > 0) {
#pragma unroll
                for (int db = 0; db < 2; ++db) { ktr[db][0] = trfrag(kprev, la, db, 0); ktr[db][1] = trfrag(kprev, la, db, 1); }
                etr[0] = etrfrag(eprev, ea, 0);
                etr[1] = etrfrag(eprev, ea, 1);
            }
            ef0 = efrag(ei, ea, 0);
            ef1 = efrag(ei, ea, 1);
#pragma unroll
            for (int s = 0; s < 4; ++s) { vfr[s] = rowfrag(vimg, la, s); kfr[s] = rowfrag(kimg, la, s); }
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
            if (!(abl & 32)) {
            if (t > 0) {
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    dq[db] = mfma(ktr[db][0], dsf0, dq[db]);
                    dq[db] = mfma(ktr[db][1], dsf1, dq[db]);
                }
                eacc = mfma(etr[0], dsf0, eacc);
                eacc = mfma(etr[1], dsf1, eacc);
            }
            sacc = mfma(ef0, as_frag(T0), zero16());
            dpacc = mfma(vfr[0], dof[0], ndl);
            sacc = mfma(ef1, as_frag(T1), sacc);
#pragma unroll
            for (int s = 1; s < 4; ++s) dpacc = mfma(vfr[s], dof[s], dpacc);
#pragma unroll
            for (int s = 0; s < 4; ++s) sacc = mfma(kfr[s], qf[s], sacc);
            } else {
                asm volatile("" : "+v"(kfr[0]), "+v"(vfr[0]), "+v"(ktr[0][0]), "+v"(etr[0]), "+v"(ef0), "+v"(ef1));
                asm volatile("" : "+v"(kfr[1]), "+v"(vfr[1]), "+v"(ktr[0][1]), "+v"(etr[1]), "+v"(kfr[2]), "+v"(kfr[3]));
                asm volatile("" : "+v"(vfr[2]), "+v"(vfr[3]), "+v"(ktr[1][0]), "+v"(ktr[1][1]));
            }
            __builtin_amdgcn_s_setprio(0);
        }
        if (t < 12) stamp(9 + 4 * t);
        if (!(abl & 128)) phase_barrier();
        if (t < 12) stamp(10 + 4 * t);
        // ---------------------------------------------------------------- V(t): write-back of tile t-1's completed key rows, staging, dS of tile t
        if (valid && t > 0 && !(abl & 16)) {
            const int ap = P == 0 ? a - 1 : a;              // period of tile t-1
            write_back(PP & 3, ap * RPP + PP);
            if constexpr (PP == PH - 1) write_back((PP + 1) & 3, ap * RPP + PP + 1);
        }
        {
            const int tw = t + 1 + grp;
            if (abl & 1) {
            } else {
            if (tw < ntile) {
                int sw = ring_next(slot);
                if (grp) sw = ring_next(sw);
                sg.store(ring + sw * STAGE_QK + soff, st);
            }
            sg.load(sbase + (size_t)min(tw + 1, ntile - 1) * 32 * ldq, ldq, st);
            }
        }
        if (valid && !(abl & 8)) {
            float ds[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) ds[r] = __builtin_amdgcn_exp2f(fmaf(sacc[r], sl, nlse2)) * dpacc[r];
            dsf0 = packfrag(ds);
            dsf1 = packfrag(ds + 8);
        }
        slot = ring_next(slot);
        if (t < 12) stamp(11 + 4 * t);
        if (!(abl & 128)) phase_barrier();
    };
    for (int a = 0; a < Hp / RPP; ++a) { A3P_FOR_PHASES(body, a) }
    stamp(2);
    // tail: second MFMA group of the last tile (phase 6 of the last period), its two completed key rows
    if (valid) {
        const unsigned char* kprev = ring + (slot == 0 ? RING - 1 : slot - 1) * STAGE_QK;
        const unsigned char* eprev = eimg + (PH - 1) * EIMG;
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            dq[db] = mfma(trfrag(kprev, la, db, 0), dsf0, dq[db]);
            dq[db] = mfma(trfrag(kprev, la, db, 1), dsf1, dq[db]);
        }
        eacc = mfma(etrfrag(eprev, ea, 0), dsf0, eacc);
        eacc = mfma(etrfrag(eprev, ea, 1), dsf1, eacc);
        write_back((PH - 1) & 3, Hp - 2);
        write_back(PH & 3, Hp - 1);
    }
    if (!grp) phase_barrier();
    __syncthreads();
    // ring + images (38 KB) become 8 private 4864-byte slices: first the fp32 kw-gradient table [32 q][28], then the dQ staging tile
    unsigned char* mine = ring + wave * 4864;
    float* twg = reinterpret_cast<float*>(mine);
    if (valid) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int rho = acc_row(reg, lane);
            if (rho < 22) twg[ql * WP + rho] = eacc[reg];
            else if (rho >= 24 && rho < 30) twg[ql * WP + rho - 2] = eacc[reg];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { dq[0][r] *= scale; dq[1][r] *= scale; }
        // r-space: dG[q][r] gathers the tables; dQ^T[d][q] += sum_r Rcat[r][d] dG[q][r]; dG is also the operand of d rel_pos
        bf16* dgrow = dG + ((size_t)(b * L + q) * H + h) * NRP;
        // the Rcat^T fragments of step s+1 are requested before the gather of step s (the loop is latency-bound otherwise: one L2 round
        // trip per step, ~1/4 of a workgroup's life at the ViT-L grid)
        bf16x8 rf[2] = {gfrag(rcatT + (size_t)ql * NRP, 0, g), gfrag(rcatT + (size_t)(32 + ql) * NRP, 0, g)};
        for (int s = 0; s < NRP / 16; ++s) {
            const int sn = min(s + 1, NRP / 16 - 1);
            const bf16x8 rn[2] = {gfrag(rcatT + (size_t)ql * NRP, sn, g), gfrag(rcatT + (size_t)(32 + ql) * NRP, sn, g)};
            float gv[8];
#pragma unroll
            for (int t8 = 0; t8 < 8; ++t8) {
                const int r = 16 * s + 8 * g + t8;
                float v = 0.f;
                if (r < 2 * Hp - 1) {
                    const int khh = qh + Hp - 1 - r;
                    if (khh >= 0 && khh < Hp) v = (float)*reinterpret_cast<const bf16*>(thT + khh * 64 + ql * 2);
                } else {
                    const int rr = r - (2 * Hp - 1);
                    const int kww = qw + WP - 1 - rr;
                    if (rr < 2 * WP - 1 && kww >= 0 && kww < WP) v = twg[ql * WP + kww];
                }
                gv[t8] = v;
            }
            const bf16x8 gf = packfrag(gv);
            *reinterpret_cast<uint4*>(dgrow + 16 * s + 8 * g) = __builtin_bit_cast(uint4, gf);
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                dq[db] = mfma(rf[db], gf, dq[db]);
                rf[db] = rn[db];
            }
        }
        stage_rows(mine, dq, 1.f, lane);                    // same-wave LDS ops are ordered: the gather above is complete
        write_rows(mine, dqkv + (size_t)(b * L + qt * 32) * ldq + h * ATT_HD, ldq, lane);
    }
    if (trace) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp(3);
}

// =============================================================================================== backward: dK, dV
// ring stage = [Q img | dO img | kw part of the T tile, chunk-swizzled [32 q][64 B] | 14 kh-table rows x [32 q] bf16 (12 key rows from the
// workgroup's first one, then -lse/scale hi, lo) | -Delta f32 [32]]
constexpr int PKV_TW = 2 * IMG, PKV_TH = PKV_TW + 2048, PKV_ND = PKV_TH + 14 * 64, PKV_STAGE = PKV_ND + 128;
__global__ __launch_bounds__(NTP, 2) void bwd_dkv_kernel(const bf16* __restrict__ qkv, size_t ldq, const bf16* __restrict__ dout, size_t lddo,
                                                         const unsigned char* __restrict__ tables, bf16* __restrict__ dqkv, int L, int H, int Hp,
                                                         float scale, int nblk, int xcd_map) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 5, ql = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), grp = wave >> 2, st = tid & 255;
    int blk, bh;
    wg_coords(nblk, xcd_map, blk, bh);
    const int b = bh / H, h = bh % H, D = H * ATT_HD;
    const bf16* qbase = qkv + (size_t)b * L * ldq + h * ATT_HD;
    const bf16* dobase = dout + (size_t)b * L * lddo + h * ATT_HD;
    const int kt = blk * NWP + wave;
    const bool valid = kt * 32 < L;
    const int key = kt * 32 + ql;
    const int kh = key / WP, kw = key % WP;
    const int khlo = (blk * NWP * 32) / WP;           // first key row of the workgroup
    const int kha = (kt * 32) / WP;                   // the wave's 32 keys lie in key rows kha and kha + 1
    const int ntile = L / 32;
    const int TB = ttile_bytes(Hp);
    const unsigned char* tbase = tables + (size_t)bh * ntile * TB;
    LaneAddr la;
    la.init(lane);
    EAddr ea;                                         // the kw part of the T tile uses the layout of the one-hot images
    ea.init(lane);
    const float sl = scale * LOG2E_F;
    const int sa = kha & 3, sb = (kha + 1) & 3;       // window slots of the wave's two key rows; the other two carry lse hi / lo

    bf16x8 kf[4], vf[4], eb0, eb1;
    int woff[2];
    {
        uint32_t e0[4] = {0, 0, 0, 0}, e1[4] = {0, 0, 0, 0};
#pragma unroll
        for (int t8 = 0; t8 < 8; ++t8) {
            const uint32_t bit = 0x3F80u << (16 * (t8 & 1));
            if (kw == 8 * g + t8) e0[t8 >> 1] |= bit;
            if (t8 < 6) {
                if (kw == 16 + 6 * g + t8) e1[t8 >> 1] |= bit;
            } else {
                const int slot4 = 2 * g + (t8 - 6);
                if (slot4 == (kh & 3) || (slot4 != sa && slot4 != sb)) e1[t8 >> 1] |= bit;
            }
        }
        eb0 = __builtin_bit_cast(bf16x8, make_uint4(e0[0], e0[1], e0[2], e0[3]));
        eb1 = __builtin_bit_cast(bf16x8, make_uint4(e1[0], e1[1], e1[2], e1[3]));
#pragma unroll
        for (int hs = 0; hs < 2; ++hs) {
            const int slot4 = 2 * g + hs;
            int row;
            if (slot4 == sa) row = kha - khlo;
            else if (slot4 == sb) row = kha + 1 - khlo;
            else {
                int rank = 0;
                for (int s2 = 0; s2 < slot4; ++s2) rank += (s2 != sa && s2 != sb) ? 1 : 0;
                row = 12 + rank;
            }
            woff[hs] = PKV_TH + row * 64 + ql * 2;
        }
    }
    if (valid) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            kf[s] = gfrag(qbase + D + (size_t)key * ldq, s, g);
            vf[s] = gfrag(qbase + 2 * D + (size_t)key * ldq, s, g);
        }
    }
    // staging: group 0 threads carry the Q tile and (first 128 of them) the kw part of the table tile, group 1 threads the dO tile and
    // (first 64) the 14 kh-table rows + -Delta
    const bf16* sbase = grp ? dobase : qbase;
    const size_t sld = grp ? lddo : ldq;
    const int soff = grp ? IMG : 0;
    Stager sg;
    uint4 rx = zero4();
    int xsrc = -1, xdst = 0;
    if (!grp) {
        if (st < 128) {
            const int qi = st >> 2, c = st & 3;
            xsrc = st * 16;
            xdst = PKV_TW + qi * 64 + ((c ^ ((qi >> 2) & 3)) << 4);
        }
    } else if (st < 56) {
        const int rs = st >> 2, c = st & 3;
        const int srow = rs < 12 ? min(khlo + rs, Hp - 1) : Hp + (rs - 12);
        xsrc = 2048 + srow * 64 + c * 16;
        xdst = PKV_TH + rs * 64 + c * 16;
    } else if (st < 64) {
        xsrc = 2048 + (Hp + 2) * 64 + (st - 56) * 16;
        xdst = PKV_ND + (st - 56) * 16;
    }
    auto load_tile = [&](int j) {
        sg.load(sbase + (size_t)j * 32 * sld, sld, st);
        if (xsrc >= 0) rx = *reinterpret_cast<const uint4*>(tbase + (size_t)j * TB + xsrc);
    };
    auto store_tile = [&](int stage) {
        unsigned char* s0 = smem + stage * PKV_STAGE;
        sg.store(s0 + soff, st);
        if (xsrc >= 0) *reinterpret_cast<uint4*>(s0 + xdst) = rx;
    };
    load_tile(0);
    store_tile(0);
    load_tile(min(1, ntile - 1));
    if (grp) {
        store_tile(1);
        load_tile(min(2, ntile - 1));
    }
    __syncthreads();
    if (grp) phase_barrier();           // group 1 runs one phase behind

    f32x16 dk[2], dv[2], sacc = zero16(), dpacc = zero16();
    dk[0] = zero16(); dk[1] = zero16(); dv[0] = zero16(); dv[1] = zero16();
    bf16x8 pf0, pf1, dsf0, dsf1;
    int slot = 0;

    for (int t = 0; t < ntile; ++t) {
        const int sprev = slot == 0 ? RING - 1 : slot - 1;
        // ---------------------------------------------------------------- M(t): dV / dK MFMAs of tile t-1, S and dP of tile t
        if (valid) {
            const unsigned char* qimg = smem + slot * PKV_STAGE;
            const unsigned char* doimg = qimg + IMG;
            const unsigned char* qprev = smem + sprev * PKV_STAGE;
            const unsigned char* doprev = qprev + IMG;
            bf16x8 dotr[2][2], qtr[2][2], qfr[4], dofr[4];
            if (t > 0) {
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    dotr[db][0] = trfrag(doprev, la, db, 0); dotr[db][1] = trfrag(doprev, la, db, 1);
                    qtr[db][0] = trfrag(qprev, la, db, 0); qtr[db][1] = trfrag(qprev, la, db, 1);
                }
            }
            const uint4 a0 = *reinterpret_cast<const uint4*>(qimg + PKV_TW + ea.row[0]);
            uint4 a1 = *reinterpret_cast<const uint4*>(qimg + PKV_TW + ea.row[1]);
            const uint32_t wlo = *reinterpret_cast<const uint16_t*>(qimg + woff[0]);
            const uint32_t whi = *reinterpret_cast<const uint16_t*>(qimg + woff[1]);
            f32x16 nd;
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const float4 n4 = *reinterpret_cast<const float4*>(qimg + PKV_ND + (8 * rg + 4 * g) * 4);
                nd[rg * 4 + 0] = n4.x; nd[rg * 4 + 1] = n4.y; nd[rg * 4 + 2] = n4.z; nd[rg * 4 + 3] = n4.w;
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) { qfr[s] = rowfrag(qimg, la, s); dofr[s] = rowfrag(doimg, la, s); }
            __builtin_amdgcn_sched_barrier(0);
            a1.w = wlo | (whi << 16);
            __builtin_amdgcn_s_setprio(1);
            if (t > 0) {
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    dv[db] = mfma(dotr[db][0], pf0, dv[db]);    // dV^T[d][key] += dO^T[d][q] P[q][key]
                    dv[db] = mfma(dotr[db][1], pf1, dv[db]);
                    dk[db] = mfma(qtr[db][0], dsf0, dk[db]);    // dK^T[d][key] += Q^T[d][q] dS[q][key]
                    dk[db] = mfma(qtr[db][1], dsf1, dk[db]);
                }
            }
            sacc = mfma(as_frag(a0), eb0, zero16());            // S[q][key] - lse: lane = key, registers = q rows
            dpacc = mfma(dofr[0], vf[0], nd);                   // dP[q][key] - Delta[q]
            sacc = mfma(as_frag(a1), eb1, sacc);
#pragma unroll
            for (int s = 1; s < 4; ++s) dpacc = mfma(dofr[s], vf[s], dpacc);
#pragma unroll
            for (int s = 0; s < 4; ++s) sacc = mfma(qfr[s], kf[s], sacc);
            __builtin_amdgcn_s_setprio(0);
        }
        phase_barrier();
        // ---------------------------------------------------------------- V(t): staging, then P and dS of tile t
        {
            const int tw = t + 1 + grp;
            if (tw < ntile) {
                int sw = ring_next(slot);
                if (grp) sw = ring_next(sw);
                store_tile(sw);
            }
            load_tile(min(tw + 1, ntile - 1));
        }
        if (valid) {
            float p[16], ds[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                p[r] = __builtin_amdgcn_exp2f(sacc[r] * sl);
                ds[r] = p[r] * dpacc[r];
            }
            pf0 = packfrag(p);
            pf1 = packfrag(p + 8);
            dsf0 = packfrag(ds);
            dsf1 = packfrag(ds + 8);
        }
        slot = ring_next(slot);
        phase_barrier();
    }
    if (valid) {
        const unsigned char* qprev = smem + (slot == 0 ? RING - 1 : slot - 1) * PKV_STAGE;
        const unsigned char* doprev = qprev + IMG;
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            dv[db] = mfma(trfrag(doprev, la, db, 0), pf0, dv[db]);
            dv[db] = mfma(trfrag(doprev, la, db, 1), pf1, dv[db]);
            dk[db] = mfma(trfrag(qprev, la, db, 0), dsf0, dk[db]);
            dk[db] = mfma(trfrag(qprev, la, db, 1), dsf1, dk[db]);
        }
    }
    if (!grp) phase_barrier();
    __syncthreads();
    unsigned char* stg = smem + wave * 2 * IMG;
    if (valid) {
        stage_rows(stg, dk, scale, lane);
        stage_rows(stg + IMG, dv, 1.f, lane);
        bf16* orow = dqkv + (size_t)(b * L + kt * 32) * ldq + h * ATT_HD;
        write_rows(stg, orow + D, ldq, lane);
        write_rows(stg + IMG, orow + 2 * D, ldq, lane);
    }
}

}   // namespace pr
}   // namespace a3

extern "C" int pa_attn_trace_paired(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(a3::pr::g_trace_p), sizeof(a3::pr::g_trace_p));
}

static int a3p_xcd_map_on() {
    static const int v = [] { const char* e = getenv("PA_ATTN_XCD"); return e ? atoi(e) : 1; }();
    return v;
}

int attn3p_fwd(const bf16* qkv, int64_t ldq, const bf16* rcat, bf16* out, int64_t ldo, float* lse, void* tables, int Bn, int L, int H,
               int Hp, int Wp, float scale, hipStream_t st) {
    using namespace a3;
    using namespace a3::pr;
    const int NRP = pa_relpos_rows_padded(Hp, Wp);
    const size_t smem = (size_t)NWP * Hp * 64 + RING * STAGE_QK + PH * EIMG;
    static bool done = false;
    if (int e = set_smem(reinterpret_cast<const void*>(fwd_kernel), done)) return e;
    const int nblk = (L / 32 + NWP - 1) / NWP;
    PA_LAUNCH(fwd_kernel, dim3(nblk * Bn * H), dim3(NTP), smem, st, qkv, (size_t)ldq, rcat, out, (size_t)ldo, lse,
              reinterpret_cast<unsigned char*>(tables), L, H, Hp, NRP, scale, nblk, a3p_xcd_map_on());
    return (int)hipGetLastError();
}

int attn3p_bwd(const bf16* qkv, int64_t ldq, const bf16* rcatT, const bf16* dout, int64_t lddo, const float* lse, const float* delta,
               void* tables, bf16* dqkv, bf16* dG, int Bn, int L, int H, int Hp, int Wp, float scale, hipStream_t st) {
    using namespace a3;
    using namespace a3::pr;
    const int NRP = pa_relpos_rows_padded(Hp, Wp);
    const int nblk = (L / 32 + NWP - 1) / NWP;
    unsigned char* tb = reinterpret_cast<unsigned char*>(tables);
    int e;
    {
        const int total = Bn * H * L;
        PA_LAUNCH(prep_kernel, dim3((total + 255) / 256), dim3(256), 0, st, lse, delta, tb, L, Hp, 1.f / scale, total);
        if ((e = (int)hipGetLastError())) return e;
    }
    {
        const size_t smem = (size_t)NWP * Hp * 64 + RING * STAGE_QK + PH * EIMG;
        static bool done = false;
        if ((e = set_smem(reinterpret_cast<const void*>(bwd_dq_kernel), done))) return e;
        const char* ablv = getenv("PA_ATTN3_ABL");           // diagnostics, read per launch
        const int abl = ablv ? atoi(ablv) : 0;
        PA_LAUNCH(bwd_dq_kernel, dim3(nblk * Bn * H), dim3(NTP), smem, st, qkv, (size_t)ldq, rcatT, dout, (size_t)lddo, lse, tb, dqkv, dG, L, H,
                  Hp, NRP, scale, nblk, a3p_xcd_map_on(), abl);
        if ((e = (int)hipGetLastError())) return e;
    }
    {
        size_t smem = (size_t)RING * PKV_STAGE;
        if (smem < (size_t)NWP * 2 * IMG) smem = (size_t)NWP * 2 * IMG;
        static bool done = false;
        if ((e = set_smem(reinterpret_cast<const void*>(bwd_dkv_kernel), done))) return e;
        PA_LAUNCH(bwd_dkv_kernel, dim3(nblk * Bn * H), dim3(NTP), smem, st, qkv, (size_t)ldq, dout, (size_t)lddo, tb, dqkv, L, H, Hp, scale, nblk,
                  a3p_xcd_map_on());
        return (int)hipGetLastError();
    }
}
