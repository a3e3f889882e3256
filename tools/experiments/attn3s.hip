// Generation-3 attention backward-dQ, software-pipelined build (4-wave workgroups, same math / tables / images as attn3.hip).
//
// s_memtime trace of attn3.hip's dQ kernel (tools/attn_trace.py, MI355X, ViT-L grid): of the ~2270 cycles a wave spends per 32x32 tile only
// ~730 are the 16 MFMAs (512) and the VALU block (220); ~400 are spent waiting for the first group's operand fragments right after
// the barrier (all waves of the CU hit the LDS at once), ~460 extra around the second MFMA group, ~370 at the staging stores.
// Here an iteration is reordered so that no MFMA waits for an LDS read issued in the same iteration:
//
//     iteration t:   16 MFMAs back to back   second group of tile t-1 (operands: packed dS / P of the previous VALU block) and first group
//                                            of tile t -- every fragment was requested during iteration t-1
//                    LDS requests            transposed fragments of tile t (for iteration t+1's second group), row fragments / one-hot /
//                                            table-window entry of tile t+1 (its first group)
//                    VALU block of tile t    exp, products, packing -- the LDS requests above are in flight meanwhile
//                    write-backs, staging    completed key rows of tile t-1 (dQ); tile t+2 registers -> ring stage, barrier
//
// K/V tiles live in a ring of three LDS stages: iteration t reads tiles t and t+1 and writes tile t+2.
// (dKV stays with attn3.hip's loop: with dK, dV, S, dP accumulators and the K, V operands resident it has no registers left for
// fragments that live across an iteration -- the pipelined form spilled inside the loop.)
#include "attn3_common.h"
#include "../../include/painter_hip.h"
#include "attn3.h"
#include <cstdlib>

namespace a3 {
namespace sp {

constexpr int RING = 3;
DEVI int ring_next(int s) { return s == RING - 1 ? 0 : s + 1; }

#define A3S_FOR_PHASES(body, a)                                                                                          \
    body(std::integral_constant<int, 0>{}, a); body(std::integral_constant<int, 1>{}, a); body(std::integral_constant<int, 2>{}, a); \
    body(std::integral_constant<int, 3>{}, a); body(std::integral_constant<int, 4>{}, a); body(std::integral_constant<int, 5>{}, a); \
    body(std::integral_constant<int, 6>{}, a);

// =============================================================================================== backward: dQ, bias gradients
// LDS: ring 3 x [K img | V img] | thT 4 waves x [Hp][32 q] bf16 (values, replaced row by row by their gradients) | 7 one-hot images
__global__ __launch_bounds__(NT, 2) void bwd_dq_kernel(const bf16* __restrict__ qkv, size_t ldq, const bf16* __restrict__ rcatT,
                                                       const bf16* __restrict__ dout, size_t lddo, const float* __restrict__ lse,
                                                       const unsigned char* __restrict__ tables, bf16* __restrict__ dqkv, bf16* __restrict__ dG,
                                                       int L, int H, int Hp, int NRP, float scale, int nblk, int xcd_map, int abl) {
    // abl (diagnostics, PA_ATTN3_ABL; results are WRONG with any bit set): 1 no staging traffic, 2 no per-tile barrier, 4 no LDS fragment
    // requests inside the loop, 8 no exp, 16 no table write-backs, 32 no second MFMA group, 64 no first MFMA group
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 5, ql = lane & 31;
    int blk, bh;
    wg_coords(nblk, xcd_map, blk, bh);
    const int b = bh / H, h = bh % H, D = H * ATT_HD;
    const bf16* base = qkv + (size_t)b * L * ldq + h * ATT_HD;
    const bf16* kbase = base + D;
    const bf16* vbase = base + 2 * D;
    const int qt = blk * NW + wave;
    const bool valid = qt * 32 < L;
    const int q = qt * 32 + ql;
    const int qh = q / WP, qw = q % WP;
    unsigned char* ring = smem;
    unsigned char* thT = smem + RING * STAGE_QK + wave * Hp * 64;
    unsigned char* eimg = smem + RING * STAGE_QK + NW * Hp * 64;
    LaneAddr la;
    la.init(lane);
    EAddr ea;
    ea.init(lane);
    const int ntile = L / 32;
    const float sl = scale * LOG2E_F;
    Stager ks, vs;
    ks.load(kbase, ldq, tid);
    vs.load(vbase, ldq, tid);
    build_eimg(eimg, tid);

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
    ks.store(ring, tid);                                                     // tile 0
    vs.store(ring + IMG, tid);
    {
        const int j1 = min(1, ntile - 1);
        ks.load(kbase + (size_t)j1 * 32 * ldq, ldq, tid);
        vs.load(vbase + (size_t)j1 * 32 * ldq, ldq, tid);
        ks.store(ring + STAGE_QK, tid);                                      // tile 1
        vs.store(ring + STAGE_QK + IMG, tid);
    }
    __syncthreads();

    f32x16 dq[2], eacc, sacc = zero16(), dpacc = zero16();
    dq[0] = zero16();
    dq[1] = zero16();
    eacc = zero16();
    unsigned char* thw = thT + ql * 2;
    // operand fragments travel from one iteration to the next
    bf16x8 ktr[2][2], etr[2], kfr[4], vfr[4], ef0, ef1, dsf0, dsf1;
    if (valid) {                        // first group of tile 0
        win_set<0>(T1.w, thw);
        win_set<1>(T1.w, thw + 64);
        ef0 = efrag(eimg, ea, 0);
        ef1 = efrag(eimg, ea, 1);
#pragma unroll
        for (int s = 0; s < 4; ++s) { vfr[s] = rowfrag(ring + IMG, la, s); kfr[s] = rowfrag(ring, la, s); }
    }
    int slot = 0;                       // ring stage of tile t

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

    auto body = [&](auto pc, int a) {
        constexpr int P = decltype(pc)::value;
        constexpr int PP = (P + PH - 1) % PH;               // phase of tile t-1
        constexpr int PN = (P + 1) % PH;                    // phase of tile t+1
        const int t = a * PH + P;
        const int s1 = ring_next(slot), s2 = ring_next(s1);
        if (!(abl & 1)) {   // tile t+2 -> registers (stored at the end of this iteration)
            const int jn = min(t + 2, ntile - 1);
            ks.load(kbase + (size_t)jn * 32 * ldq, ldq, tid);
            vs.load(vbase + (size_t)jn * 32 * ldq, ldq, tid);
        }
        if (valid) {
            // ---- 16 MFMAs: second group of tile t-1, first group of tile t
            __builtin_amdgcn_sched_barrier(0);
            if (t > 0 && !(abl & 32)) {
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    dq[db] = mfma(ktr[db][0], dsf0, dq[db]);
                    dq[db] = mfma(ktr[db][1], dsf1, dq[db]);
                }
                eacc = mfma(etr[0], dsf0, eacc);
                eacc = mfma(etr[1], dsf1, eacc);
            }
            if (!(abl & 64)) {
            sacc = mfma(ef0, as_frag(T0), zero16());
            dpacc = mfma(vfr[0], dof[0], ndl);
            sacc = mfma(ef1, as_frag(T1), sacc);
#pragma unroll
            for (int s = 1; s < 4; ++s) dpacc = mfma(vfr[s], dof[s], dpacc);
#pragma unroll
            for (int s = 0; s < 4; ++s) sacc = mfma(kfr[s], qf[s], sacc);
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- LDS requests for the next iteration
            const unsigned char* kimg = ring + slot * STAGE_QK;
            if (!(abl & 4)) {
#pragma unroll
            for (int db = 0; db < 2; ++db) { ktr[db][0] = trfrag(kimg, la, db, 0); ktr[db][1] = trfrag(kimg, la, db, 1); }
            etr[0] = etrfrag(eimg + P * EIMG, ea, 0);
            etr[1] = etrfrag(eimg + P * EIMG, ea, 1);
            if (t + 1 < ntile) {
                const unsigned char* kn = ring + s1 * STAGE_QK;
                const unsigned char* thr = thw + (P == PH - 1 ? a + 1 : a) * (RPP * 64);      // period of tile t+1
                if constexpr (PN == 0) win_set<0>(T1.w, thr);
                win_set<(PN + 1) & 1>(T1.w, thr + (PN + 1) * 64);
                ef0 = efrag(eimg + PN * EIMG, ea, 0);
                ef1 = efrag(eimg + PN * EIMG, ea, 1);
#pragma unroll
                for (int s = 0; s < 4; ++s) { vfr[s] = rowfrag(kn + IMG, la, s); kfr[s] = rowfrag(kn, la, s); }
            }
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- VALU block of tile t
            float ds[16];
            if (abl & 8) {
#pragma unroll
                for (int r = 0; r < 16; ++r) ds[r] = fmaf(sacc[r], sl, nlse2) * dpacc[r];
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) ds[r] = __builtin_amdgcn_exp2f(fmaf(sacc[r], sl, nlse2)) * dpacc[r];
            }
            dsf0 = packfrag(ds);
            dsf1 = packfrag(ds + 8);
            // ---- completed key rows of tile t-1
            if (t > 0 && !(abl & 16)) {
                const int ap = P == 0 ? a - 1 : a;
                write_back(PP & 3, ap * RPP + PP);
                if constexpr (PP == PH - 1) write_back((PP + 1) & 3, ap * RPP + PP + 1);
            }
        }
        if (t + 2 < ntile && !(abl & 1)) {
            ks.store(ring + s2 * STAGE_QK, tid);
            vs.store(ring + s2 * STAGE_QK + IMG, tid);
        }
        slot = s1;
        if (!(abl & 2)) __syncthreads();
    };
    for (int a = 0; a < Hp / RPP; ++a) { A3S_FOR_PHASES(body, a) }
    // tail: second group of the last tile (phase 6 of the last period), its two completed key rows
    if (valid) {
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            dq[db] = mfma(ktr[db][0], dsf0, dq[db]);
            dq[db] = mfma(ktr[db][1], dsf1, dq[db]);
        }
        eacc = mfma(etr[0], dsf0, eacc);
        eacc = mfma(etr[1], dsf1, eacc);
        write_back((PH - 1) & 3, Hp - 2);
        write_back(PH & 3, Hp - 1);
    }
    __syncthreads();
    // ring + images become 4 private slices: the fp32 kw-gradient table [32 q][28] (images region), the dQ staging tile (ring region)
    float* twg = reinterpret_cast<float*>(eimg + wave * (32 * WP * 4));
    unsigned char* stg = ring + wave * IMG;
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
        stage_rows(stg, dq, 1.f, lane);
        write_rows(stg, dqkv + (size_t)(b * L + qt * 32) * ldq + h * ATT_HD, ldq, lane);
    }
}

}   // namespace sp
}   // namespace a3

static int a3s_xcd_map_on() {
    static const int v = [] { const char* e = getenv("PA_ATTN_XCD"); return e ? atoi(e) : 1; }();
    return v;
}

int attn3s_dq(const bf16* qkv, int64_t ldq, const bf16* rcatT, const bf16* dout, int64_t lddo, const float* lse, void* tables, bf16* dqkv,
              bf16* dG, int Bn, int L, int H, int Hp, int Wp, float scale, hipStream_t st) {
    using namespace a3;
    using namespace a3::sp;
    const int NRP = pa_relpos_rows_padded(Hp, Wp);
    const int nblk = (L / 32 + NW - 1) / NW;
    const size_t smem = (size_t)RING * STAGE_QK + (size_t)NW * Hp * 64 + PH * EIMG;
    static bool done = false;
    if (int e = set_smem(reinterpret_cast<const void*>(bwd_dq_kernel), done)) return e;
    const char* ab = getenv("PA_ATTN3_ABL");          // diagnostics: read per call so that one process can sweep it
    PA_LAUNCH(bwd_dq_kernel, dim3(nblk * Bn * H), dim3(NT), smem, st, qkv, (size_t)ldq, rcatT, dout, (size_t)lddo, lse,
              reinterpret_cast<const unsigned char*>(tables), dqkv, dG, L, H, Hp, NRP, scale, nblk, a3s_xcd_map_on(), ab ? atoi(ab) : 0);
    return (int)hipGetLastError();
}
