// bf16 fused attention with decomposed rel-pos bias, second generation (forward, backward-dQ, backward-dKV).
// Same math and same C ABI as attn_fwd.hip / attn_bwd.hip (Painter/models_painter.py:76-86, util/vitdet_utils.py:63-125,
// SURVEY.md 8a a5-a8, a17, Appendix B.2); those files keep the exact-fp32 build and the shapes this file does not cover.
//
// What changed against generation 1, and why (rocprof: 1 workgroup of 4-7 waves per CU, 96-270 TFLOP/s):
//   * LDS per query row drops from 672 B (two fp32 k-space tables) to 224 B: the kw table stays fp32 (it is loaded
//     straight into the S accumulator as the MFMA's C operand -- no VALU add), the kh table is bf16 (one scalar per run of
//     4 keys), and the bias GRADIENT needs no table at all (next point).  3 workgroups of 4 waves fit a CU (4 for the
//     single-stage forward kernel, the default).
//   * d bias / d (kw), d bias / d (kh) are contractions of dS with one-hot key patterns, so they run on the matrix pipe:
//     E^T[32][32 keys] . dS^T accumulates rows 0..Wp-1 = dGw[q][kw] over all key tiles and rows 28..31 = this tile's
//     dGh[q][kh0..kh0+3]; the latter slide through a 4-deep register window (key rows are visited in order) and replace
//     the consumed entries of the kh table in place.  No LDS atomics.
//   * one LDS image per K / V / Q / dO tile: natural [row][64 d] order, XOR-swizzled so that BOTH the row-fragment
//     ds_read_b128 (contraction over d) and the transposing ds_read_b64_tr_b16 (contraction over the tile's rows) are
//     bank-conflict free.  No transposed copies, no register transposes.
//   * forward: the running max is only re-based when a tile exceeds it by more than 2^6 (wave-uniform, rare branch), so
//     the usual O rescale and the subtraction of the new max disappear from the per-tile VALU stream.
// Work split (all three): workgroup = 4 waves, wave = 32 rows (queries, or keys in dKV), lane = one row end to end;
// 32-row tiles of the other axis stream through LDS, register-staged (global -> VGPR early, VGPR -> LDS late); STAGES = 2
// double-buffers them (one barrier per tile), STAGES = 1 trades the second stage for one more resident workgroup.
#include "attn_tile_hd.h"
#include "../../include/painter_hip.h"
#include "attn2.h"
#include <cstdlib>

namespace a2 {
using namespace atile;

constexpr float THR = 6.0f;
// run table: for tile phase ph, half-wave g, run rg: low 16 bits = kw * 4 (byte offset into the lane's kw table row),
// high 16 bits = (kh - kh0(tile)) * 2 (byte offset into the kh table row, relative to the tile's first key row)
DEVI void build_rtab(uint32_t* rtab, int nphase, int Wp, int tid) {
    if (tid < nphase * 8) {
        const int ph = tid >> 3, g = (tid >> 2) & 1, rg = tid & 3;
        const int a = (32 * ph) % Wp + 8 * rg + 4 * g;
        rtab[tid] = (uint32_t)((a % Wp) * 4) | ((uint32_t)((a / Wp) * 2) << 16);
    }
}
DEVI float bf_lo(uint32_t w) { return __builtin_bit_cast(float, w << 16); }
DEVI float bf_hi(uint32_t w) { return __builtin_bit_cast(float, w & 0xffff0000u); }

// k-space bias tables of this lane's query row:  tw[kw] = (q . rel_pos_w[qw - kw + Wp-1]) / scale   (fp32; C operand of S)
//                                                th[kh] = (q . rel_pos_h[qh - kh + Hp-1]) * log2 e  (bf16)
template <int HD>
DEVI void build_tables(float* tw, bf16* th, const bf16* rcat, int NRP, const bf16x8 (&qf)[HD / 16], int qh, int qw, int Hp, int Wp,
                       float inv_scale, int lane) {
    const int g = lane >> 5;
    for (int rbk = 0; rbk < NRP / 32; ++rbk) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const bf16* rp = rcat + (size_t)(rbk * 32 + (lane & 31)) * HD;
#pragma unroll
        for (int s = 0; s < HD / 16; ++s) acc = mfma(gfrag(rp, s, g), qf[s], acc);
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int r = rbk * 32 + acc_row(reg, lane);
            if (r < 2 * Hp - 1) {
                const int kh = qh + Hp - 1 - r;
                if (kh >= 0 && kh < Hp) th[kh] = (bf16)(acc[reg] * LOG2E_F);
            } else {
                const int rr = r - (2 * Hp - 1);
                const int kw = qw + Wp - 1 - rr;
                if (rr < 2 * Wp - 1 && kw >= 0 && kw < Wp) tw[kw] = acc[reg] * inv_scale;
            }
        }
    }
}

// =============================================================================================== forward
// LDS: [K img | V img] x 2 stages (16 KB) | tw f32 [128][Wp] | th bf16 [128][thld] | run table [nphase][2][4] u32
// STAGES = 2: K/V double-buffered in LDS, one barrier per key tile, 3 workgroups per CU.  STAGES = 1: one K/V stage (two barriers
// per tile), 38 KB of LDS -> 4 workgroups per CU (needs <= 128 VGPRs: PF = 1).
template <int PF, int STAGES, int HD = ATT_HD>
__global__ __launch_bounds__(NT, STAGES == 1 ? (HD == 64 ? 4 : 3) : 2) void fwd_kernel(const bf16* __restrict__ qkv, size_t ldq, const bf16* __restrict__ rcat,
                                                  bf16* __restrict__ out, size_t ldo, float* __restrict__ lse, int L, int H, int Hp,
                                                  int Wp, int NRP, float scale, int thld, int nphase, int nblk, int xcd_map) {
    typedef TileOps<HD> TO;
    constexpr int KS = TO::KS, DB = TO::DB, IMGB = TO::IMG_B, STQK = 2 * TO::IMG_B;
    typedef typename TO::Stager Stager;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 5, row = wave * 32 + (lane & 31);
    int blk, bh;
    wg_coords(nblk, xcd_map, blk, bh);
    const int b = bh / H, h = bh % H, D = H * HD;
    const bf16* base = qkv + (size_t)b * L * ldq + h * HD;
    const bf16* kbase = base + D;
    const bf16* vbase = base + 2 * D;
    const int qt = blk * NW + wave;
    const bool valid = qt * 32 < L;
    const int q = qt * 32 + (lane & 31);
    unsigned char* twb = smem + STAGES * STQK + (size_t)row * Wp * 4;
    unsigned char* thb = smem + STAGES * STQK + ROWS * Wp * 4 + (size_t)row * thld * 2;
    uint32_t* rtab = reinterpret_cast<uint32_t*>(smem + STAGES * STQK + ROWS * Wp * 4 + ROWS * thld * 2);
    LaneAddr la;
    la.init(lane);
    build_rtab(rtab, nphase, Wp, tid);

    bf16x8 qf[KS];
    if (valid) {
#pragma unroll
        for (int s = 0; s < KS; ++s) qf[s] = gfrag(base + (size_t)q * ldq, s, g);
        build_tables<HD>(reinterpret_cast<float*>(twb), reinterpret_cast<bf16*>(thb), rcat, NRP, qf, q / Wp, q % Wp, Hp, Wp, 1.f / scale, lane);
    }
    // K/V tiles are register-staged PF tiles ahead (global -> VGPR at the top of iteration j for tile j + PF, VGPR -> LDS at the
    // bottom of iteration j for tile j + 1): with PF = 2 a load has a whole iteration of MFMA work to cover its L2 latency.
    Stager ksA, vsA, ksB, vsB;
    const int ntile = L / 32;
    ksA.load(kbase, ldq, tid);
    vsA.load(vbase, ldq, tid);
    ksA.store(smem, tid);
    vsA.store(smem + IMGB, tid);
    if (PF == 2) {
        const int j1 = min(1, ntile - 1);
        ksA.load(kbase + (size_t)j1 * 32 * ldq, ldq, tid);
        vsA.load(vbase + (size_t)j1 * 32 * ldq, ldq, tid);
    }
    __syncthreads();

    f32x16 oacc[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[db][r] = 0.f;
    float m = 0.f, l = 0.f;
    const float sl = scale * LOG2E_F;
    int phase = 0;

    auto iter = [&](int j, Stager& kl, Stager& vl, Stager& kst, Stager& vst) {
        {   // unconditional (clamped to the last tile): a static number of loads in flight lets hipcc emit a counted vmcnt wait
            // for the set that is stored below instead of draining the loads just issued
            const int jn = min(j + PF, ntile - 1);
            kl.load(kbase + (size_t)jn * 32 * ldq, ldq, tid);
            vl.load(vbase + (size_t)jn * 32 * ldq, ldq, tid);
        }
        const unsigned char* kimg = smem + (STAGES == 2 ? (j & 1) * STQK : 0);
        const unsigned char* vimg = kimg + IMGB;
        if (valid) {
            const uint4 rt = *reinterpret_cast<const uint4*>(rtab + (phase * 2 + g) * 4);
            const uint32_t rts[4] = {rt.x, rt.y, rt.z, rt.w};
            const unsigned char* tht = thb + ((32 * j) / Wp) * 2;
            f32x16 sacc;
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const float4 bw = *reinterpret_cast<const float4*>(twb + (rts[rg] & 0xffffu));
                sacc[rg * 4 + 0] = bw.x; sacc[rg * 4 + 1] = bw.y; sacc[rg * 4 + 2] = bw.z; sacc[rg * 4 + 3] = bw.w;
            }
#pragma unroll
            for (int s = 0; s < KS; ++s) sacc = mfma(TO::rowfrag(kimg, la, s, lane), qf[s], sacc);
            float p[16];
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const float bhm = (float)*reinterpret_cast<const bf16*>(tht + (rts[rg] >> 16)) - m;
#pragma unroll
                for (int e = 0; e < 4; ++e) p[rg * 4 + e] = fmaf(sacc[rg * 4 + e], sl, bhm);
            }
            float tmax = max16(p);
            tmax = fmaxf(tmax, xor32(tmax));
            if (j == 0 || __any(tmax > THR)) {          // wave-uniform; after the first tiles almost never taken
                const float delta = (j == 0) ? tmax : fmaxf(tmax, 0.f);
                const float alpha = __builtin_amdgcn_exp2f(-delta);
                m += delta;
                l *= alpha;
#pragma unroll
                for (int r = 0; r < 16; ++r) p[r] -= delta;
#pragma unroll
                for (int db = 0; db < DB; ++db)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) p[r] = __builtin_amdgcn_exp2f(p[r]);
            l += sum16(p);
            const bf16x8 pf0 = packfrag(p), pf1 = packfrag(p + 8);
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                oacc[db] = mfma(TO::trfrag(vimg, la, db, 0, lane), pf0, oacc[db]);
                oacc[db] = mfma(TO::trfrag(vimg, la, db, 1, lane), pf1, oacc[db]);
            }
        }
        phase = phase + 1 == nphase ? 0 : phase + 1;
        if constexpr (STAGES == 1) __syncthreads();      // every wave has finished reading the only stage
        if (j + 1 < ntile) {
            kst.store(smem + (STAGES == 2 ? ((j + 1) & 1) * STQK : 0), tid);
            vst.store(smem + (STAGES == 2 ? ((j + 1) & 1) * STQK : 0) + IMGB, tid);
        }
        __syncthreads();
    };
    if constexpr (PF == 1) {
        for (int j = 0; j < ntile; ++j) iter(j, ksA, vsA, ksA, vsA);
    } else {
        int j = 0;
        for (; j + 1 < ntile; j += 2) {
            iter(j, ksB, vsB, ksA, vsA);
            iter(j + 1, ksA, vsA, ksB, vsB);
        }
        if (j < ntile) iter(j, ksB, vsB, ksA, vsA);
    }
    // the K/V stages are free now: per-wave 4 KB staging tile
    unsigned char* stg = smem + wave * TO::STG_B;
    if (valid) {
        const float lt = l + xor32(l);
        if (g == 0) lse[(size_t)bh * L + q] = (m + __builtin_amdgcn_logf(lt)) * LN2_F;
        TO::stage_rows(stg, oacc, 1.f / lt, lane);
        TO::write_rows(stg, out + (size_t)(b * L + qt * 32) * ldo + h * HD, ldo, lane);   // same-wave LDS ops are ordered
    }
}

// =============================================================================================== backward: one-hot key patterns
// etab[phase][s][g][row 32][8 slots] bf16: E^T[row][key] for the tile whose first key is 32*phase (mod Wp-periodic):
//   row < Wp: 1 iff kw(key) == row;   row = 28 + i: 1 iff kh(key) - kh(first key of the tile) == i
__global__ void etab_kernel(bf16* etab, int Wp, int nphase) {
    const int phase = blockIdx.x;
    const int off = (32 * phase) % Wp;
    for (int idx = threadIdx.x; idx < 1024; idx += blockDim.x) {
        const int slot = idx & 7, rrow = (idx >> 3) & 31, g = (idx >> 8) & 1, s = idx >> 9;
        const int key = 16 * s + 4 * g + (slot & 3) + 8 * (slot >> 2);
        const int a = off + key, kw = a % Wp, dk = a / Wp;
        const bool one = rrow < Wp ? (kw == rrow) : (rrow >= 28 && dk == rrow - 28);
        etab[(size_t)phase * 1024 + idx] = (bf16)(one ? 1.f : 0.f);
    }
}
static int gcd_i(int a, int b) { return b ? gcd_i(b, a % b) : a; }
static int etab_phases(int Wp) { return Wp / gcd_i(32, Wp); }
constexpr int ETAB_BYTES = 32 * 2048;     // up to 32 phases

// aux tile of (bh, q-tile), consumed by the dKV kernel with 16-byte reads:
//   tabhT f32 [Hp][32] (the bf16-rounded values the other kernels use) | tabwT f32 [Wp][32] = tw[q][kw] - lse2[q] / (scale log2 e)
//   | -Delta f32 [32]
DEVI size_t aux_tile_bytes(int Hp, int Wp) { return (size_t)(Hp + Wp) * 128 + 128; }

// =============================================================================================== backward: dQ, bias gradients
// STAGES as in the forward kernel: one K/V stage brings the LDS footprint under 53 KB, i.e. 3 workgroups per CU (with MINW = 3)
// WP32 (key rows of 32 tokens = exactly one 32-key tile, ViT-H/14): the one-hot pattern of the kw gradient is the identity and a tile holds
// ONE key row, so E^T . dS^T is dS^T itself (16 adds into the same accumulator layout) and the kh gradient of the tile is the row sum of dS
template <int MINW, int STAGES, int HD = ATT_HD, bool WP32 = false>
__global__ __launch_bounds__(NT, MINW) void bwd_dq_kernel(const bf16* __restrict__ qkv, size_t ldq, const bf16* __restrict__ rcat,
                                                     const bf16* __restrict__ rcatT, const bf16* __restrict__ dout, size_t lddo,
                                                     const float* __restrict__ lse, const float* __restrict__ delta,
                                                     bf16* __restrict__ dqkv, bf16* __restrict__ dG, unsigned char* __restrict__ aux,
                                                     int L, int H, int Hp, int Wp, int NRP, float scale, int thld, int nphase, int nblk,
                                                     int xcd_map) {
    typedef TileOps<HD> TO;
    constexpr int KS = TO::KS, DB = TO::DB, IMGB = TO::IMG_B, STQK = 2 * TO::IMG_B;
    typedef typename TO::Stager Stager;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 5, row = wave * 32 + (lane & 31);
    int blk, bh;
    wg_coords(nblk, xcd_map, blk, bh);
    const int b = bh / H, h = bh % H, D = H * HD;
    const bf16* base = qkv + (size_t)b * L * ldq + h * HD;
    const bf16* kbase = base + D;
    const bf16* vbase = base + 2 * D;
    const int qt = blk * NW + wave;
    const bool valid = qt * 32 < L;
    const int q = qt * 32 + (lane & 31);
    const int qh = q / Wp, qw = q % Wp;
    unsigned char* twb = smem + STAGES * STQK + (size_t)row * Wp * 4;
    unsigned char* thb = smem + STAGES * STQK + ROWS * Wp * 4 + (size_t)row * thld * 2;
    float* tw = reinterpret_cast<float*>(twb);
    bf16* th = reinterpret_cast<bf16*>(thb);
    uint32_t* rtab = reinterpret_cast<uint32_t*>(smem + STAGES * STQK + ROWS * Wp * 4 + ROWS * thld * 2);
    // one-hot key patterns of every tile phase, copied once into LDS (a per-tile global load would sit on the critical path)
    unsigned char* etab = reinterpret_cast<unsigned char*>(rtab) + 1024;
    for (int c = tid; c < nphase * 128; c += NT)
        *reinterpret_cast<uint4*>(etab + c * 16) = *reinterpret_cast<const uint4*>(aux + (size_t)c * 16);
    LaneAddr la;
    la.init(lane);
    build_rtab(rtab, nphase, Wp, tid);
    const float sl = scale * LOG2E_F;

    bf16x8 qf[KS], dof[KS];
    float lse2 = 0.f;
    f32x16 ndl;              // -Delta[q] in every register: C operand of the first dP MFMA, so dpacc = dP - Delta for free
#pragma unroll
    for (int r = 0; r < 16; ++r) ndl[r] = 0.f;
    if (valid) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            qf[s] = gfrag(base + (size_t)q * ldq, s, g);
            dof[s] = gfrag(dout + (size_t)(b * L + q) * lddo + h * HD, s, g);
        }
        lse2 = lse[(size_t)bh * L + q] * LOG2E_F;
        const float ndlt = -delta[(size_t)bh * L + q];
#pragma unroll
        for (int r = 0; r < 16; ++r) ndl[r] = ndlt;
        build_tables<HD>(tw, th, rcat, NRP, qf, qh, qw, Hp, Wp, 1.f / scale, lane);
        // export the transposed tables for the dKV kernel (own-wave LDS writes above are ordered before these reads)
        unsigned char* at = aux + ETAB_BYTES + ((size_t)bh * (L / 32) + qt) * aux_tile_bytes(Hp, Wp);
        float* ahT = reinterpret_cast<float*>(at) + (lane & 31);
        float* awT = reinterpret_cast<float*>(at + (size_t)Hp * 128) + (lane & 31);
        const float ls = lse2 / sl;
        for (int c = g; c < Hp; c += 2) ahT[(size_t)c * 32] = (float)th[c];
        for (int c = g; c < Wp; c += 2) awT[(size_t)c * 32] = tw[c] - ls;
        if (g == 0) reinterpret_cast<float*>(at + (size_t)(Hp + Wp) * 128)[lane & 31] = ndlt;
    }
    Stager ks, vs;
    const int ntile = L / 32;
    ks.load(kbase, ldq, tid);
    vs.load(vbase, ldq, tid);
    ks.store(smem, tid);
    vs.store(smem + IMGB, tid);
    __syncthreads();

    f32x16 dq[DB], eacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) eacc[r] = 0.f;
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[db][r] = 0.f;
    float wh[4] = {0.f, 0.f, 0.f, 0.f};
    int kh0 = 0, phase = 0;

    for (int j = 0; j < ntile; ++j) {
        if (j + 1 < ntile) {
            ks.load(kbase + (size_t)(j + 1) * 32 * ldq, ldq, tid);
            vs.load(vbase + (size_t)(j + 1) * 32 * ldq, ldq, tid);
        }
        const unsigned char* kimg = smem + (STAGES == 2 ? (j & 1) * STQK : 0);
        const unsigned char* vimg = kimg + IMGB;
        const int kh0n = (32 * (j + 1)) / Wp;
        if (valid) {
            const unsigned char* ep = etab + phase * 2048 + g * 512 + (lane & 31) * 16;
            bf16x8 ef0, ef1;
            if constexpr (!WP32) {
                ef0 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(ep));
                ef1 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(ep + 1024));
            }
            const uint4 rt = *reinterpret_cast<const uint4*>(rtab + (phase * 2 + g) * 4);
            const uint32_t rts[4] = {rt.x, rt.y, rt.z, rt.w};
            const unsigned char* tht = thb + kh0 * 2;
            f32x16 sacc, dpacc;
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const float4 bw = *reinterpret_cast<const float4*>(twb + (rts[rg] & 0xffffu));
                sacc[rg * 4 + 0] = bw.x; sacc[rg * 4 + 1] = bw.y; sacc[rg * 4 + 2] = bw.z; sacc[rg * 4 + 3] = bw.w;
            }
            sacc = mfma(TO::rowfrag(kimg, la, 0, lane), qf[0], sacc);
            dpacc = mfma(TO::rowfrag(vimg, la, 0, lane), dof[0], ndl);
#pragma unroll
            for (int s = 1; s < KS; ++s) {
                sacc = mfma(TO::rowfrag(kimg, la, s, lane), qf[s], sacc);
                dpacc = mfma(TO::rowfrag(vimg, la, s, lane), dof[s], dpacc);
            }
            float ds[16];
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const float bhx = (float)*reinterpret_cast<const bf16*>(tht + (rts[rg] >> 16)) - lse2;
#pragma unroll
                for (int e = 0; e < 4; ++e) ds[rg * 4 + e] = __builtin_amdgcn_exp2f(fmaf(sacc[rg * 4 + e], sl, bhx));
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) ds[r] *= dpacc[r];
            const bf16x8 dsf0 = packfrag(ds), dsf1 = packfrag(ds + 8);
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                dq[db] = mfma(TO::trfrag(kimg, la, db, 0, lane), dsf0, dq[db]);
                dq[db] = mfma(TO::trfrag(kimg, la, db, 1, lane), dsf1, dq[db]);
            }
            if constexpr (WP32) {
                // kw gradient: the tile's keys are kw = 0..31 in the accumulator's own row order; kh gradient: this key row's sum
#pragma unroll
                for (int r = 0; r < 16; ++r) eacc[r] += ds[r];
                const float hs = sum16(ds);
                const float ht = hs + xor32(hs);
                if (g && kh0 < Hp) th[kh0] = (bf16)ht;
            } else {
            eacc = mfma(ef0, dsf0, eacc);
            eacc = mfma(ef1, dsf1, eacc);
            }
            // rows 28..31 of eacc (registers 12..15 of the upper half-wave) are this tile's dGh[q][kh0 .. kh0+3]
            if (!WP32 && g) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { wh[i] += eacc[12 + i]; eacc[12 + i] = 0.f; }
                const int dlt = kh0n - kh0;             // 1..3 key rows completed by this tile (wave-uniform)
                if (dlt >= 1 && kh0 < Hp) th[kh0] = (bf16)wh[0];
                if (dlt >= 2 && kh0 + 1 < Hp) th[kh0 + 1] = (bf16)wh[1];
                if (dlt >= 3 && kh0 + 2 < Hp) th[kh0 + 2] = (bf16)wh[2];
                if (dlt == 1) { wh[0] = wh[1]; wh[1] = wh[2]; wh[2] = wh[3]; wh[3] = 0.f; }
                else if (dlt == 2) { wh[0] = wh[2]; wh[1] = wh[3]; wh[2] = 0.f; wh[3] = 0.f; }
                else if (dlt >= 3) { wh[0] = wh[3]; wh[1] = 0.f; wh[2] = 0.f; wh[3] = 0.f; }
            }
        }
        kh0 = kh0n;
        phase = phase + 1 == nphase ? 0 : phase + 1;
        if constexpr (STAGES == 1) __syncthreads();      // every wave has finished reading the only stage
        if (j + 1 < ntile) {
            ks.store(smem + (STAGES == 2 ? ((j + 1) & 1) * STQK : 0), tid);
            vs.store(smem + (STAGES == 2 ? ((j + 1) & 1) * STQK : 0) + IMGB, tid);
        }
        __syncthreads();
    }

    unsigned char* stg = smem + wave * TO::STG_B;
    if (valid) {
        // the tables now become the k-space bias gradients: th[kh] = dGh (already, entry by entry), tw[kw] = dGw
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int r = acc_row(reg, lane);
            if (r < Wp) tw[r] = eacc[reg];
        }
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) dq[db][r] *= scale;
        // r-space: dG[q][r] gathers the tables; dQ^T[d][q] += sum_r Rcat[r][d] dG[q][r]; dG is also the operand of d rel_pos
        bf16* dgrow = dG + ((size_t)(b * L + q) * H + h) * NRP;
        for (int s = 0; s < NRP / 16; ++s) {
            float gv[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int r = 16 * s + 8 * g + t;
                float v = 0.f;
                if (r < 2 * Hp - 1) {
                    const int khh = qh + Hp - 1 - r;
                    if (khh >= 0 && khh < Hp) v = (float)th[khh];
                } else {
                    const int rr = r - (2 * Hp - 1);
                    const int kww = qw + Wp - 1 - rr;
                    if (rr < 2 * Wp - 1 && kww >= 0 && kww < Wp) v = tw[kww];
                }
                gv[t] = v;
            }
            const bf16x8 gf = packfrag(gv);
            *reinterpret_cast<uint4*>(dgrow + 16 * s + 8 * g) = __builtin_bit_cast(uint4, gf);
#pragma unroll
            for (int db = 0; db < DB; ++db)
                dq[db] = mfma(TO::gtfrag(rcatT, (size_t)NRP, db, s, lane), gf, dq[db]);
        }
    }
    // single K/V stage: the staging tiles of waves 2, 3 overlap other waves' bias tables, which the loop above still reads
    if constexpr (STAGES == 1) __syncthreads();
    if (valid) {
        TO::stage_rows(stg, dq, 1.f, lane);
        TO::write_rows(stg, dqkv + (size_t)(b * L + qt * 32) * ldq + h * HD, ldq, lane);
    }
}

// =============================================================================================== backward: dK, dV
constexpr int AH_LD = 144, AW_LD = 144;    // LDS row strides (bytes) of the transposed kh / kw tables (32 f32 + pad)
template <int MINW, int HD = ATT_HD>
__global__ __launch_bounds__(NT, MINW) void bwd_dkv_kernel(const bf16* __restrict__ qkv, size_t ldq, const bf16* __restrict__ dout,
                                                      size_t lddo, const unsigned char* __restrict__ aux, bf16* __restrict__ dqkv,
                                                      int L, int H, int Hp, int Wp, float scale, int nblk, int xcd_map) {
    typedef TileOps<HD> TO;
    constexpr int KS = TO::KS, DB = TO::DB, IMGB = TO::IMG_B, STQK = 2 * TO::IMG_B;
    typedef typename TO::Stager Stager;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 5;
    int blk, bh;
    wg_coords(nblk, xcd_map, blk, bh);
    const int b = bh / H, h = bh % H, D = H * HD;
    const bf16* qbase = qkv + (size_t)b * L * ldq + h * HD;
    const bf16* dobase = dout + (size_t)b * L * lddo + h * HD;
    const int kt = blk * NW + wave;
    const bool valid = kt * 32 < L;
    const int key = kt * 32 + (lane & 31);
    const int khl = key / Wp, kwl = key % Wp;
    const int stage_bytes = STQK + Hp * AH_LD + Wp * AW_LD + 128;
    const int n_h = Hp * 8, n_w = Wp * 8, n_aux = n_h + n_w + 8;       // 16-byte chunks of the aux tile
    const size_t atb = aux_tile_bytes(Hp, Wp);
    const unsigned char* aux_bh = aux + ETAB_BYTES + (size_t)bh * (L / 32) * atb;
    LaneAddr la;
    la.init(lane);
    const float sl = scale * LOG2E_F;

    bf16x8 kf[KS], vf[KS];
    if (valid) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            kf[s] = gfrag(qbase + D + (size_t)key * ldq, s, g);
            vf[s] = gfrag(qbase + 2 * D + (size_t)key * ldq, s, g);
        }
    }
    Stager qs, dos;
    uint4 ra0, ra1, ra2, ra3;      // aux tile chunks (named registers: an indexed array would live in scratch)
    const int ntile = L / 32;
    auto aux_store = [&](unsigned char* ah, unsigned char* aw, int c, const uint4& v) {
        if (c < n_h) *reinterpret_cast<uint4*>(ah + (c >> 3) * AH_LD + (c & 7) * 16) = v;
        else if (c < n_h + n_w) *reinterpret_cast<uint4*>(aw + ((c - n_h) >> 3) * AW_LD + ((c - n_h) & 7) * 16) = v;
        else if (c < n_aux) *reinterpret_cast<uint4*>(aw + Wp * AW_LD + (c - n_h - n_w) * 16) = v;
    };
    auto load_all = [&](int j) {
        qs.load(qbase + (size_t)j * 32 * ldq, ldq, tid);
        dos.load(dobase + (size_t)j * 32 * lddo, lddo, tid);
        const unsigned char* at = aux_bh + (size_t)j * atb;
        if (tid < n_aux) ra0 = *reinterpret_cast<const uint4*>(at + (size_t)tid * 16);
        if (tid + NT < n_aux) ra1 = *reinterpret_cast<const uint4*>(at + (size_t)(tid + NT) * 16);
        if (tid + 2 * NT < n_aux) ra2 = *reinterpret_cast<const uint4*>(at + (size_t)(tid + 2 * NT) * 16);
        if (tid + 3 * NT < n_aux) ra3 = *reinterpret_cast<const uint4*>(at + (size_t)(tid + 3 * NT) * 16);
    };
    auto store_all = [&](int stage) {
        unsigned char* s0 = smem + stage * stage_bytes;
        qs.store(s0, tid);
        dos.store(s0 + IMGB, tid);
        unsigned char* ah = s0 + STQK;
        unsigned char* aw = ah + Hp * AH_LD;
        aux_store(ah, aw, tid, ra0);
        aux_store(ah, aw, tid + NT, ra1);
        aux_store(ah, aw, tid + 2 * NT, ra2);
        aux_store(ah, aw, tid + 3 * NT, ra3);
    };
    load_all(0);
    store_all(0);
    __syncthreads();

    f32x16 dk[DB], dv[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[db][r] = 0.f; dv[db][r] = 0.f; }

    for (int j = 0; j < ntile; ++j) {
        if (j + 1 < ntile) load_all(j + 1);
        const unsigned char* qimg = smem + (j & 1) * stage_bytes;
        const unsigned char* doimg = qimg + IMGB;
        if (valid) {
            const unsigned char* ah = qimg + STQK + khl * AH_LD;
            const unsigned char* aw = qimg + STQK + Hp * AH_LD + kwl * AW_LD;
            const unsigned char* ad = qimg + STQK + Hp * AH_LD + Wp * AW_LD;
            f32x16 sacc, dpacc;
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int q0 = 8 * rg + 4 * g;
                const float4 bw = *reinterpret_cast<const float4*>(aw + q0 * 4);
                const float4 nd = *reinterpret_cast<const float4*>(ad + q0 * 4);
                sacc[rg * 4 + 0] = bw.x; sacc[rg * 4 + 1] = bw.y; sacc[rg * 4 + 2] = bw.z; sacc[rg * 4 + 3] = bw.w;
                dpacc[rg * 4 + 0] = nd.x; dpacc[rg * 4 + 1] = nd.y; dpacc[rg * 4 + 2] = nd.z; dpacc[rg * 4 + 3] = nd.w;
            }
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                sacc = mfma(TO::rowfrag(qimg, la, s, lane), kf[s], sacc);          // S[q][key]: lane = key, registers = q rows
                dpacc = mfma(TO::rowfrag(doimg, la, s, lane), vf[s], dpacc);       // dP[q][key] - Delta[q]
            }
            float p[16], ds[16];
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const float4 b4 = *reinterpret_cast<const float4*>(ah + (8 * rg + 4 * g) * 4);
                p[rg * 4 + 0] = __builtin_amdgcn_exp2f(fmaf(sacc[rg * 4 + 0], sl, b4.x));
                p[rg * 4 + 1] = __builtin_amdgcn_exp2f(fmaf(sacc[rg * 4 + 1], sl, b4.y));
                p[rg * 4 + 2] = __builtin_amdgcn_exp2f(fmaf(sacc[rg * 4 + 2], sl, b4.z));
                p[rg * 4 + 3] = __builtin_amdgcn_exp2f(fmaf(sacc[rg * 4 + 3], sl, b4.w));
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) ds[r] = p[r] * dpacc[r];
            const bf16x8 pf0 = packfrag(p), pf1 = packfrag(p + 8), dsf0 = packfrag(ds), dsf1 = packfrag(ds + 8);
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                dv[db] = mfma(TO::trfrag(doimg, la, db, 0, lane), pf0, dv[db]);    // dV^T[d][key] += dO^T[d][q] P[q][key]
                dv[db] = mfma(TO::trfrag(doimg, la, db, 1, lane), pf1, dv[db]);
                dk[db] = mfma(TO::trfrag(qimg, la, db, 0, lane), dsf0, dk[db]);    // dK^T[d][key] += Q^T[d][q] dS[q][key]
                dk[db] = mfma(TO::trfrag(qimg, la, db, 1, lane), dsf1, dk[db]);
            }
        }
        if (j + 1 < ntile) store_all((j + 1) & 1);
        __syncthreads();
    }
    unsigned char* stg = smem + wave * 2 * TO::STG_B;
    if (valid) {
        TO::stage_rows(stg, dk, scale, lane);
        TO::stage_rows(stg + TO::STG_B, dv, 1.f, lane);
        bf16* orow = dqkv + (size_t)(b * L + kt * 32) * ldq + h * HD;
        TO::write_rows(stg, orow + D, ldq, lane);
        TO::write_rows(stg + TO::STG_B, orow + 2 * D, ldq, lane);
    }
}

}   // namespace a2

// head_dim 64: key rows of 12..28 tokens (the 28-wide grid normally runs on generation 3); head_dim 80 (round 4, ViT-H/14): the same
// plus key rows of exactly 32 tokens (WP32: the 64 x 32 grid of 896 x 448 at patch 14)
bool attn2_ok(int L, int Hp, int Wp, int hd) {
    if (!(L == Hp * Wp && L % 32 == 0 && Wp % 4 == 0 && Wp >= 12 && Hp % 2 == 0 && Hp >= 2)) return false;
    if (hd == 64) return Wp <= 28;
    if (hd == 80) return (Wp <= 28 || Wp == 32) && Hp * 8 + Wp * 8 + 8 <= 4 * a2::NT;
    return false;
}

// bf16 row stride of the kh table: an odd number of dwords, so the 32 rows of a wave fall into 32 different banks
static int th_ld(int Hp) {
    int s = Hp + 2;
    if (((s / 2) & 1) == 0) s += 2;
    return s;
}
static int xcd_map_on() {
    static const int v = [] { const char* e = getenv("PA_ATTN_XCD"); return e ? atoi(e) : 1; }();
    return v;
}
template <int HD> static size_t qside_smem(int Hp, int Wp) {
    return 2 * 2 * (size_t)a2::TileOps<HD>::IMG_B + (size_t)a2::ROWS * (Wp * 4 + th_ld(Hp) * 2) + 1024;
}

template <int HD>
static int attn2_fwd_t(const bf16* qkv, int64_t ldq, const bf16* rcat, bf16* out, int64_t ldo, float* lse, int Bn, int L, int H, int Hp, int Wp,
                       float scale, hipStream_t st) {
    using namespace a2;
    typedef TileOps<HD> TO;
    const int NRP = pa_relpos_rows_padded(Hp, Wp);
    size_t smem = qside_smem<HD>(Hp, Wp);
    auto smem_bytes_single_stage = [&](size_t& b) { b -= 2 * TO::IMG_B; if (b < (size_t)NW * TO::STG_B) b = (size_t)NW * TO::STG_B; };
    // PA_ATTN_FWD_PF: 0 (default) = single LDS stage, 4 workgroups per CU (measured +12 % over the double-buffered forms: the kernel
    // is latency/issue-bound and the fourth wave per SIMD buys more than the second barrier costs); 1 = one-tile prefetch,
    // double-buffered; 2 = two-tile prefetch, double-buffered
    static const int pf = [] { const char* v = getenv("PA_ATTN_FWD_PF"); return v ? atoi(v) : 0; }();
    static bool done0 = false, done1 = false, done2 = false;
    auto kern = pf == 0 ? fwd_kernel<1, 1, HD> : (pf == 1 ? fwd_kernel<1, 2, HD> : fwd_kernel<2, 2, HD>);
    if (pf == 0) smem_bytes_single_stage(smem);
    if (int e = set_smem(reinterpret_cast<const void*>(kern), pf == 0 ? done0 : (pf == 1 ? done1 : done2))) return e;
    const int nblk = (L / 32 + NW - 1) / NW;
    PA_LAUNCH(kern, dim3(nblk * Bn * H), dim3(NT), smem, st, qkv, (size_t)ldq, rcat, out, (size_t)ldo, lse, L, H, Hp,
              Wp, NRP, scale, th_ld(Hp), etab_phases(Wp), nblk, xcd_map_on());
    return (int)hipGetLastError();
}
int attn2_fwd(const bf16* qkv, int64_t ldq, const bf16* rcat, bf16* out, int64_t ldo, float* lse, int Bn, int L, int H, int Hp, int Wp,
              int hd, float scale, hipStream_t st) {
    if (hd == 80) return attn2_fwd_t<80>(qkv, ldq, rcat, out, ldo, lse, Bn, L, H, Hp, Wp, scale, st);
    return attn2_fwd_t<64>(qkv, ldq, rcat, out, ldo, lse, Bn, L, H, Hp, Wp, scale, st);
}

int64_t attn2_aux_bytes(int Bn, int L, int H, int Hp, int Wp) {
    return (int64_t)a2::ETAB_BYTES + (int64_t)Bn * H * (L / 32) * ((int64_t)(Hp + Wp) * 128 + 128);
}

template <int HD, bool WP32>
static int attn2_bwd_t(const bf16* qkv, int64_t ldq, const bf16* rcat, const bf16* rcatT, const bf16* dout, int64_t lddo, const float* lse,
                       const float* delta, bf16* dqkv, bf16* dG, void* aux, int Bn, int L, int H, int Hp, int Wp, float scale, hipStream_t st) {
    using namespace a2;
    typedef TileOps<HD> TO;
    const int NRP = pa_relpos_rows_padded(Hp, Wp);
    const int nphase = etab_phases(Wp);
    PA_LAUNCH(etab_kernel, dim3(nphase), dim3(256), 0, st, reinterpret_cast<bf16*>(aux), Wp, nphase);
    int e = (int)hipGetLastError();
    if (e) return e;
    // PA_ATTN_DQ_WAVES / PA_ATTN_DKV_WAVES: 2 (default) or 3 waves per SIMD; dQ with 3 also uses the single K/V stage (head_dim 64 only)
    static const int dq_w = [] { const char* v = getenv("PA_ATTN_DQ_WAVES"); return v ? atoi(v) : 2; }();
    static const int minw = [] { const char* v = getenv("PA_ATTN_DKV_WAVES"); return v ? atoi(v) : 2; }();
    {
        size_t smem = qside_smem<HD>(Hp, Wp) + (size_t)nphase * 2048;
        const bool w3 = dq_w == 3 && HD == 64 && !WP32;
        if (w3) smem -= 2 * TO::IMG_B;
        auto kern = w3 ? bwd_dq_kernel<3, 1, HD, WP32> : bwd_dq_kernel<2, 2, HD, WP32>;
        static bool done2 = false, done3 = false;
        if ((e = set_smem(reinterpret_cast<const void*>(kern), w3 ? done3 : done2))) return e;
        const int nblk = (L / 32 + NW - 1) / NW;
        PA_LAUNCH(kern, dim3(nblk * Bn * H), dim3(NT), smem, st, qkv, (size_t)ldq, rcat, rcatT, dout, (size_t)lddo,
                  lse, delta, dqkv, dG, reinterpret_cast<unsigned char*>(aux), L, H, Hp, Wp, NRP, scale, th_ld(Hp), nphase, nblk, xcd_map_on());
        if ((e = (int)hipGetLastError())) return e;
    }
    {
        size_t smem = 2 * (size_t)(2 * TO::IMG_B + Hp * AH_LD + Wp * AW_LD + 128);
        if (smem < (size_t)NW * 2 * TO::STG_B) smem = (size_t)NW * 2 * TO::STG_B;
        if (Hp * 8 + Wp * 8 + 8 > 4 * NT) return (int)hipErrorInvalidValue;
        const bool w3 = minw == 3 && HD == 64;
        auto kern = w3 ? bwd_dkv_kernel<3, HD> : bwd_dkv_kernel<2, HD>;
        static bool done2 = false, done3 = false;
        if ((e = set_smem(reinterpret_cast<const void*>(kern), w3 ? done3 : done2))) return e;
        const int nblk = (L / 32 + NW - 1) / NW;
        PA_LAUNCH(kern, dim3(nblk * Bn * H), dim3(NT), smem, st, qkv, (size_t)ldq, dout, (size_t)lddo,
                  reinterpret_cast<const unsigned char*>(aux), dqkv, L, H, Hp, Wp, scale, nblk, xcd_map_on());
        return (int)hipGetLastError();
    }
}
int attn2_bwd(const bf16* qkv, int64_t ldq, const bf16* rcat, const bf16* rcatT, const bf16* dout, int64_t lddo, const float* lse,
              const float* delta, bf16* dqkv, bf16* dG, void* aux, int Bn, int L, int H, int Hp, int Wp, int hd, float scale, hipStream_t st) {
#define A2_BWD(HD_, W32_) attn2_bwd_t<HD_, W32_>(qkv, ldq, rcat, rcatT, dout, lddo, lse, delta, dqkv, dG, aux, Bn, L, H, Hp, Wp, scale, st)
    if (hd == 80) return Wp == 32 ? A2_BWD(80, true) : A2_BWD(80, false);
    return A2_BWD(64, false);
#undef A2_BWD
}
