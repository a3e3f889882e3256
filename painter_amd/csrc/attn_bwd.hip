// Fused attention backward (autograd of Painter/models_painter.py:76-86 + util/vitdet_utils.py:96-125;
// SURVEY.md 8a a17, Appendix B.2).  P is recomputed from the saved row log-sum-exp; nothing L x L is stored.
//
//   Delta[q] = sum_d dO[q,d] O[q,d]                                 (pa_attn_bwd_delta)
//   dS = P o (dP - Delta),  dP = dO V^T
//   kernel A (lane = query row, loops over key tiles):
//       dQ = scale dS K + dG Rcat            dG[q][r] = r-space scatter of the k-space bias-gradient tables
//       tabg[q][kh] += sum_kw dS,  tabg[q][Hp+kw] += sum_kh dS      (LDS: ds_add_f32 / in-order RMW)
//       exports  aux[bh][qtile][TS+2][32] = transposed bias table (x log2 e), lse*log2 e, Delta
//                dG[R][H][NRP] (T) for the rel_pos_h / rel_pos_w weight gradient (a TN contraction on the GEMM engine)
//   kernel B (lane = key, loops over query tiles):
//       dV = P^T dO,   dK = scale dS^T Q        (bias / lse / Delta come from `aux`, 16-byte reads)
#include "attn_common.h"
#include <cstdlib>
#include "gemm_engine.h"
#include "../../include/painter_hip.h"
#include "attn2.h"
#include "attn3.h"

// ------------------------------------------------------------------------------- Delta pre-pass
// one wave per row; 16 elements per lane.  head_dim 64: a group of 4 consecutive lanes covers one head (DPP combine).  Any other
// head_dim % 16 == 0 (80: ViT-H/14, whose attention has been on a timed path since round 4): the head_dim / 16 partial sums of a head are
// combined in ascending order by one lane through LDS.  (Until the end of round 5 that branch was "one lane per head": 16 of 64 lanes active
// with 80 scalar loads each -- 96 us per launch for 42 MB, 3.1 ms of the ViT-H/14 step; profiles/r05_vit_huge_one_stream_kernel_stats.csv.)
template <typename T> DEVI void load16(const T* p, float* v);           // 16 consecutive elements from a 16-byte aligned address
template <> DEVI void load16<bf16>(const bf16* p, float* v) {
    const uint4 a = *reinterpret_cast<const uint4*>(p), b = *reinterpret_cast<const uint4*>(p + 8);
    v[0] = bf16_lo(a.x); v[1] = bf16_hi(a.x); v[2] = bf16_lo(a.y); v[3] = bf16_hi(a.y);
    v[4] = bf16_lo(a.z); v[5] = bf16_hi(a.z); v[6] = bf16_lo(a.w); v[7] = bf16_hi(a.w);
    v[8] = bf16_lo(b.x); v[9] = bf16_hi(b.x); v[10] = bf16_lo(b.y); v[11] = bf16_hi(b.y);
    v[12] = bf16_lo(b.z); v[13] = bf16_hi(b.z); v[14] = bf16_lo(b.w); v[15] = bf16_hi(b.w);
}
template <> DEVI void load16<float>(const float* p, float* v) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 a = *reinterpret_cast<const float4*>(p + 4 * i);
        v[4 * i] = a.x; v[4 * i + 1] = a.y; v[4 * i + 2] = a.z; v[4 * i + 3] = a.w;
    }
}
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void attn_delta_kernel(const T* o, size_t ldo, const T* d_o, size_t lddo, float* delta, int R, int L, int H, int hd) {
    __shared__ float part[4][128];              // head_dim != 64: one partial sum per 16-element chunk of the wave's row (D <= 2048)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    const bool live = row < R;                  // (no early return: the general branch has a workgroup barrier)
    const int b = live ? row / L : 0, l = live ? row % L : 0;
    const int D = H * hd;
    if (hd == 64) {
        if (!live) return;
        for (int c = lane * 16; c < D; c += 1024) {
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) s += to_f(o[(size_t)row * ldo + c + e]) * to_f(d_o[(size_t)row * lddo + c + e]);
            s = quad_sum(s);
            if ((lane & 3) == 0) delta[((size_t)b * H + c / 64) * L + l] = s;
        }
    } else {
        const int nchunk = D / 16, gl = hd / 16;
        if (live) {
            for (int c = lane; c < nchunk; c += 64) {
                float s = 0.f;
                const T* po = o + (size_t)row * ldo + c * 16;
                const T* pd = d_o + (size_t)row * lddo + c * 16;
                if constexpr (VEC) {
                    float a[16], g[16];
                    load16<T>(po, a);
                    load16<T>(pd, g);
#pragma unroll
                    for (int e = 0; e < 16; ++e) s += a[e] * g[e];
                } else {
#pragma unroll
                    for (int e = 0; e < 16; ++e) s += to_f(po[e]) * to_f(pd[e]);
                }
                part[wave][c] = s;
            }
        }
        __syncthreads();
        if (live) {
            for (int h = lane; h < H; h += 64) {
                float s = 0.f;
                for (int k = 0; k < gl; ++k) s += part[wave][h * gl + k];
                delta[((size_t)b * H + h) * L + l] = s;
            }
        }
    }
}
extern "C" int pa_attn_bwd_delta(int dtype, const void* out, int64_t ldo, const void* dout, int64_t lddo, float* delta, int batch, int L,
                                 int heads, int head_dim, hipStream_t st) {
    if (head_dim <= 0 || head_dim % 16) return (int)hipErrorInvalidValue;
    if (head_dim != 64 && heads * head_dim > 2048) return (int)hipErrorInvalidValue;
    const int R = batch * L;
    const size_t esz = dtype == PA_BF16 ? 2 : 4;
    // 16-byte loads where both operands allow them (base address and row stride): always the case for the engine's tensors
    const bool vec = ((uintptr_t)out % 16 == 0) && ((uintptr_t)dout % 16 == 0) && ((size_t)ldo * esz % 16 == 0) && ((size_t)lddo * esz % 16 == 0);
#define PA_DELTA(T_, V_) PA_LAUNCH((attn_delta_kernel<T_, V_>), dim3((R + 3) / 4), dim3(256), 0, st, (const T_*)out, (size_t)ldo, (const T_*)dout, (size_t)lddo, delta, R, L, heads, head_dim)
    if (dtype == PA_BF16) { if (vec) PA_DELTA(bf16, true); else PA_DELTA(bf16, false); }
    else { if (vec) PA_DELTA(float, true); else PA_DELTA(float, false); }
#undef PA_DELTA
    LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------- kernel A: dQ + bias-gradient tables
template <typename T, int NW, int HD>
__global__ __launch_bounds__(NW * 64) void attn_bwd_dq_kernel(const T* __restrict__ qkv, size_t ldq, const T* __restrict__ rcat,
                                                              const T* __restrict__ rcatT, const T* __restrict__ dout, size_t lddo,
                                                              const float* __restrict__ lse, const float* __restrict__ delta,
                                                              T* __restrict__ dqkv, T* __restrict__ dG, float* __restrict__ aux, int L,
                                                              int H, int Hp, int Wp, int NRP, float scale, int tab_stride) {
    typedef KvTile<T, HD> KV;
    constexpr int NT = NW * 64;
    constexpr int KB = KV::KB, VB = KV::VB, KS = KV::KS, DB = KV::DB;
    constexpr int STG = 2 * KB + VB;           // one stage: K row image, V row image, K^T image
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 5;
    const int bh = blockIdx.y, b = bh / H, h = bh % H;
    const int D = H * HD, TS = Hp + Wp;
    const T* base = qkv + (size_t)b * L * ldq + h * HD;
    const T* kbase = base + D;
    const T* vbase = base + 2 * D;
    const int qt = blockIdx.x * NW + wave;
    const bool valid = qt * 32 < L;
    const int q = qt * 32 + (lane & 31);
    const int qh = q / Wp, qw = q % Wp;
    // LDS: stage s in {0,1}: K row image, V row image, K^T image; then per-wave [bias table | grad table]
    unsigned char* wreg = smem + 2 * STG + (size_t)wave * tab_stride;
    KV::zero_pad(smem + 2 * KB, tid, NT);
    KV::zero_pad(smem + STG + 2 * KB, tid, NT);
    float* tab = reinterpret_cast<float*>(wreg) + (lane & 31) * TS;
    float* tabg = reinterpret_cast<float*>(wreg) + 32 * TS + (lane & 31) * TS;

    Frag<T> qf[KS], dof[KS];
    float lse2 = 0.f, dlt = 0.f;
    if (valid) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            load_gfrag<T>(qf[s], base + (size_t)q * ldq, s, g);
            load_gfrag<T>(dof[s], dout + (size_t)(b * L + q) * lddo + h * HD, s, g);
        }
        lse2 = lse[(size_t)bh * L + q] * LOG2E_F;
        dlt = delta[(size_t)bh * L + q];
        build_bias_table<T, HD>(tab, rcat, NRP, qf, qh, qw, Hp, Wp, lane);
        for (int c = g; c < TS; c += 2) tabg[c] = 0.f;
        // export the transposed table + per-row scalars for kernel B
        float* ax = aux + ((size_t)bh * (L / 32) + qt) * (TS + 2) * 32 + (lane & 31);
        for (int c = g; c < TS; c += 2) ax[(size_t)c * 32] = tab[c];      // own-wave LDS writes above are in order
        if (g == 0) ax[(size_t)TS * 32] = lse2;
        else ax[(size_t)(TS + 1) * 32] = dlt;
    }

    RowStage<T, NT, HD> ks, vs;
    TrStage<T, HD> kts;
    const int ntile = L / 32;
    ks.load(kbase, ldq, tid);
    kts.load(kbase, ldq, tid);
    vs.load(vbase, ldq, tid);
    ks.store(smem, tid);
    vs.store(smem + KB, tid);
    kts.store(smem + 2 * KB, tid);
    __syncthreads();

    f32x16 dq[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[db][r] = 0.f;
    const float sl = scale * LOG2E_F;
    int kh[4], kw[4];
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
        const int ks0 = 8 * rg + 4 * g;
        kh[rg] = ks0 / Wp;
        kw[rg] = ks0 % Wp;
    }
    const int dq_ = 32 / Wp, dr_ = 32 % Wp;

    for (int j = 0; j < ntile; ++j) {
        if (j + 1 < ntile) {
            const T* kp = kbase + (size_t)(j + 1) * 32 * ldq;
            ks.load(kp, ldq, tid);
            kts.load(kp, ldq, tid);
            vs.load(vbase + (size_t)(j + 1) * 32 * ldq, ldq, tid);
        }
        const unsigned char* st = smem + (j & 1) * STG;
        if (valid) {
            f32x16 sacc, dpacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; dpacc[r] = 0.f; }
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                Frag<T> kf, vf;
                load_rowfrag<T, HD>(kf, st, lane & 31, s, g);
                load_rowfrag<T, HD>(vf, st + KB, lane & 31, s, g);
                mma(sacc, kf, qf[s]);
                mma(dpacc, vf, dof[s]);
            }
            float ds[16];
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const float bhv = tab[kh[rg]];
                const float4 bw = *reinterpret_cast<const float4*>(tab + Hp + kw[rg]);
                const float p0 = __builtin_amdgcn_exp2f(fmaf(sacc[rg * 4 + 0], sl, bhv + bw.x) - lse2);
                const float p1 = __builtin_amdgcn_exp2f(fmaf(sacc[rg * 4 + 1], sl, bhv + bw.y) - lse2);
                const float p2 = __builtin_amdgcn_exp2f(fmaf(sacc[rg * 4 + 2], sl, bhv + bw.z) - lse2);
                const float p3 = __builtin_amdgcn_exp2f(fmaf(sacc[rg * 4 + 3], sl, bhv + bw.w) - lse2);
                ds[rg * 4 + 0] = p0 * (dpacc[rg * 4 + 0] - dlt);
                ds[rg * 4 + 1] = p1 * (dpacc[rg * 4 + 1] - dlt);
                ds[rg * 4 + 2] = p2 * (dpacc[rg * 4 + 2] - dlt);
                ds[rg * 4 + 3] = p3 * (dpacc[rg * 4 + 3] - dlt);
                atomicAdd(tabg + kh[rg], (ds[rg * 4] + ds[rg * 4 + 1]) + (ds[rg * 4 + 2] + ds[rg * 4 + 3]));
                if (Wp >= 8) {      // the two half-waves of a query own disjoint kw runs (kw0 and kw0 + 4): in-order RMW
                    float4* gw = reinterpret_cast<float4*>(tabg + Hp + kw[rg]);
                    float4 cur = *gw;
                    cur.x += ds[rg * 4]; cur.y += ds[rg * 4 + 1]; cur.z += ds[rg * 4 + 2]; cur.w += ds[rg * 4 + 3];
                    *gw = cur;
                } else {            // Wp == 4: both halves hit kw 0..3 in the same instruction
#pragma unroll
                    for (int e = 0; e < 4; ++e) atomicAdd(tabg + Hp + kw[rg] + e, ds[rg * 4 + e]);
                }
            }
            Frag<T> dsf[2];
            pack_frag<T>(dsf[0], ds);
            pack_frag<T>(dsf[1], ds + 8);
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    Frag<T> ktf;
                    load_trfrag<T, HD>(ktf, st + 2 * KB, db * 32 + (lane & 31), s, g);
                    mma(dq[db], ktf, dsf[s]);
                }
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                kh[rg] += dq_;
                kw[rg] += dr_;
                if (kw[rg] >= Wp) { kw[rg] -= Wp; kh[rg] += 1; }
            }
        }
        if (j + 1 < ntile) {
            unsigned char* sn = smem + ((j + 1) & 1) * STG;
            ks.store(sn, tid);
            vs.store(sn + KB, tid);
            kts.store(sn + 2 * KB, tid);
        }
        __syncthreads();
    }

    // bias part of dQ through r-space: dQ^T[d][q] = scale * acc + sum_r Rcat[r][d] dG[q][r]; also emit dG (T)
    constexpr int ROWB = HD * sizeof(T);
    if (valid) {
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) dq[db][r] *= scale;
        T* dgrow = dG + ((size_t)(b * L + q) * H + h) * NRP;
        for (int s = 0; s < NRP / 16; ++s) {
            float gv[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int r = 16 * s + 8 * g + t;
                float v = 0.f;
                if (r < 2 * Hp - 1) {
                    const int khh = qh + Hp - 1 - r;
                    if (khh >= 0 && khh < Hp) v = tabg[khh];
                } else {
                    const int rr = r - (2 * Hp - 1);
                    const int kww = qw + Wp - 1 - rr;
                    if (rr < 2 * Wp - 1 && kww >= 0 && kww < Wp) v = tabg[Hp + kww];
                }
                gv[t] = v;
            }
            Frag<T> gf;
            pack_frag<T>(gf, gv);
            if constexpr (sizeof(T) == 2) {
                *reinterpret_cast<uint4*>(dgrow + 16 * s + 8 * g) = __builtin_bit_cast(uint4, gf.v);
            } else {
                *reinterpret_cast<float4*>(dgrow + 16 * s + 8 * g) = make_float4(gv[0], gv[1], gv[2], gv[3]);
                *reinterpret_cast<float4*>(dgrow + 16 * s + 8 * g + 4) = make_float4(gv[4], gv[5], gv[6], gv[7]);
            }
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                Frag<T> rf;
                const int dr = db * 32 + (lane & 31);                                  // rows d >= HD of the last block: zero operand
                const T* rp = rcatT + (size_t)(dr < HD ? dr : 0) * NRP + 16 * s + 8 * g;
                if constexpr (sizeof(T) == 2) rf.set(dr < HD ? *reinterpret_cast<const uint4*>(rp) : zero4());
                else rf.set(dr < HD ? *reinterpret_cast<const uint4*>(rp) : zero4(), dr < HD ? *reinterpret_cast<const uint4*>(rp + 4) : zero4());
                mma(dq[db], rf, gf);
            }
        }
    }
    __syncthreads();      // every wave is done with its tables -> reuse the region as the dQ staging tile
    if (valid) {
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int d0 = db * 32 + 8 * rg + 4 * g;
                if (d0 < HD) {
                    T* dst = reinterpret_cast<T*>(wreg + (lane & 31) * ROWB) + d0;
                    *reinterpret_cast<typename TT<T>::Vec4*>(dst) =
                        cvt4(dq[db][rg * 4], dq[db][rg * 4 + 1], dq[db][rg * 4 + 2], dq[db][rg * 4 + 3], (T*)nullptr);
                }
            }
    }
    __syncthreads();
    if (valid) {
        constexpr int CPR = ROWB / 16;
#pragma unroll
        for (int i = 0; i < 32 * CPR / 64; ++i) {
            const int c = lane + 64 * i, row = c / CPR, ch = c % CPR;
            const uint4 v = *reinterpret_cast<const uint4*>(wreg + row * ROWB + ch * 16);
            *reinterpret_cast<uint4*>(dqkv + (size_t)(b * L + qt * 32 + row) * ldq + h * HD + ch * TT<T>::EPC) = v;
        }
    }
}

// ------------------------------------------------------------------------------- kernel B: dK, dV
#define AUX_LD 36   // floats per LDS row of the transposed aux tile (32 + 4 pad: conflict-free 16-B reads)
template <typename T, int NW, int HD>
__global__ __launch_bounds__(NW * 64) void attn_bwd_dkv_kernel(const T* __restrict__ qkv, size_t ldq, const T* __restrict__ dout,
                                                               size_t lddo, const float* __restrict__ aux, T* __restrict__ dqkv, int L,
                                                               int H, int Hp, int Wp, float scale) {
    typedef KvTile<T, HD> KV;
    constexpr int NT = NW * 64;
    constexpr int KB = KV::KB, VB = KV::VB, KS = KV::KS, DB = KV::DB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 5;
    const int bh = blockIdx.y, b = bh / H, h = bh % H;
    const int D = H * HD, TS = Hp + Wp;
    const int AUXB = (TS + 2) * AUX_LD * 4;
    const int STG = 2 * KB + 2 * VB + AUXB;    // Q row, dO row, Q^T, dO^T, aux
    const T* qbase = qkv + (size_t)b * L * ldq + h * HD;
    const T* dobase = dout + (size_t)b * L * lddo + h * HD;
    const int kt = blockIdx.x * NW + wave;
    const bool valid = kt * 32 < L;
    const int key = kt * 32 + (lane & 31);
    const int khl = key / Wp, kwl = key % Wp;

    Frag<T> kf[KS], vf[KS];
    if (valid) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            load_gfrag<T>(kf[s], qbase + D + (size_t)key * ldq, s, g);
            load_gfrag<T>(vf[s], qbase + 2 * D + (size_t)key * ldq, s, g);
        }
    }
    RowStage<T, NT, HD> qs, dos;
    TrStage<T, HD> qts, dots;
    const int NAUX = (TS + 2) * 8;                       // 16-B chunks of the aux tile
    constexpr int CAUX = HD == 64 ? 3 : 4;               // up to CAUX * NT chunks: TS + 2 <= 96 (HD 64: the 56 x 28 grid) / 128 (ViT-H/14: 64 x 32 -> 98)
    uint4 ra[CAUX];
    const int ntile = L / 32;
    auto load_all = [&](int j) {
        qs.load(qbase + (size_t)j * 32 * ldq, ldq, tid);
        qts.load(qbase + (size_t)j * 32 * ldq, ldq, tid);
        dos.load(dobase + (size_t)j * 32 * lddo, lddo, tid);
        dots.load(dobase + (size_t)j * 32 * lddo, lddo, tid);
        const float* ax = aux + ((size_t)bh * ntile + j) * (TS + 2) * 32;
#pragma unroll
        for (int i = 0; i < CAUX; ++i) {
            const int c = tid + NT * i;
            if (c < NAUX) ra[i] = *reinterpret_cast<const uint4*>(ax + (size_t)c * 4);
        }
    };
    auto store_all = [&](int stage) {
        unsigned char* s0 = smem + stage * STG;
        qs.store(s0, tid);
        dos.store(s0 + KB, tid);
        qts.store(s0 + 2 * KB, tid);
        dots.store(s0 + 2 * KB + VB, tid);
#pragma unroll
        for (int i = 0; i < CAUX; ++i) {
            const int c = tid + NT * i;
            if (c < NAUX) *reinterpret_cast<uint4*>(s0 + 2 * KB + 2 * VB + (c >> 3) * (AUX_LD * 4) + (c & 7) * 16) = ra[i];
        }
    };
    for (int stage = 0; stage < 2; ++stage) {
        KV::zero_pad(smem + stage * STG + 2 * KB, tid, NT);
        KV::zero_pad(smem + stage * STG + 2 * KB + VB, tid, NT);
    }
    load_all(0);
    store_all(0);
    __syncthreads();

    f32x16 dk[DB], dv[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[db][r] = 0.f; dv[db][r] = 0.f; }
    const float sl = scale * LOG2E_F;

    for (int j = 0; j < ntile; ++j) {
        if (j + 1 < ntile) load_all(j + 1);
        const unsigned char* st = smem + (j & 1) * STG;
        if (valid) {
            f32x16 sacc, dpacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; dpacc[r] = 0.f; }
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                Frag<T> qf, dof;
                load_rowfrag<T, HD>(qf, st, lane & 31, s, g);
                load_rowfrag<T, HD>(dof, st + KB, lane & 31, s, g);
                mma(sacc, qf, kf[s]);          // S[q][key]: lane = key, regs = q rows
                mma(dpacc, dof, vf[s]);        // dP[q][key]
            }
            const float* ax = reinterpret_cast<const float*>(st + 2 * KB + 2 * VB);
            float p[16], ds[16];
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int q0 = 8 * rg + 4 * g;
                const float4 bh4 = *reinterpret_cast<const float4*>(ax + khl * AUX_LD + q0);
                const float4 bw4 = *reinterpret_cast<const float4*>(ax + (Hp + kwl) * AUX_LD + q0);
                const float4 ls4 = *reinterpret_cast<const float4*>(ax + TS * AUX_LD + q0);
                const float4 dl4 = *reinterpret_cast<const float4*>(ax + (TS + 1) * AUX_LD + q0);
                p[rg * 4 + 0] = __builtin_amdgcn_exp2f(fmaf(sacc[rg * 4 + 0], sl, bh4.x + bw4.x) - ls4.x);
                p[rg * 4 + 1] = __builtin_amdgcn_exp2f(fmaf(sacc[rg * 4 + 1], sl, bh4.y + bw4.y) - ls4.y);
                p[rg * 4 + 2] = __builtin_amdgcn_exp2f(fmaf(sacc[rg * 4 + 2], sl, bh4.z + bw4.z) - ls4.z);
                p[rg * 4 + 3] = __builtin_amdgcn_exp2f(fmaf(sacc[rg * 4 + 3], sl, bh4.w + bw4.w) - ls4.w);
                ds[rg * 4 + 0] = p[rg * 4 + 0] * (dpacc[rg * 4 + 0] - dl4.x);
                ds[rg * 4 + 1] = p[rg * 4 + 1] * (dpacc[rg * 4 + 1] - dl4.y);
                ds[rg * 4 + 2] = p[rg * 4 + 2] * (dpacc[rg * 4 + 2] - dl4.z);
                ds[rg * 4 + 3] = p[rg * 4 + 3] * (dpacc[rg * 4 + 3] - dl4.w);
            }
            Frag<T> pf[2], dsf[2];
            pack_frag<T>(pf[0], p);
            pack_frag<T>(pf[1], p + 8);
            pack_frag<T>(dsf[0], ds);
            pack_frag<T>(dsf[1], ds + 8);
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    Frag<T> qtf, dotf;
                    load_trfrag<T, HD>(dotf, st + 2 * KB + VB, db * 32 + (lane & 31), s, g);
                    load_trfrag<T, HD>(qtf, st + 2 * KB, db * 32 + (lane & 31), s, g);
                    mma(dv[db], dotf, pf[s]);      // dV^T[d][key] += dO^T[d][q] P[q][key]
                    mma(dk[db], qtf, dsf[s]);      // dK^T[d][key] += Q^T[d][q] dS[q][key]
                }
        }
        if (j + 1 < ntile) store_all((j + 1) & 1);
        __syncthreads();
    }

    // store dK (x scale) and dV rows through a per-wave LDS staging tile
    constexpr int ROWB = HD * sizeof(T);
    unsigned char* stg = smem + (size_t)wave * 2 * 32 * ROWB;
    if (valid) {
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int d0 = db * 32 + 8 * rg + 4 * g;
                if (d0 >= HD) continue;
                T* dstk = reinterpret_cast<T*>(stg + (lane & 31) * ROWB) + d0;
                T* dstv = reinterpret_cast<T*>(stg + 32 * ROWB + (lane & 31) * ROWB) + d0;
                *reinterpret_cast<typename TT<T>::Vec4*>(dstk) = cvt4(dk[db][rg * 4] * scale, dk[db][rg * 4 + 1] * scale,
                                                                    dk[db][rg * 4 + 2] * scale, dk[db][rg * 4 + 3] * scale, (T*)nullptr);
                *reinterpret_cast<typename TT<T>::Vec4*>(dstv) =
                    cvt4(dv[db][rg * 4], dv[db][rg * 4 + 1], dv[db][rg * 4 + 2], dv[db][rg * 4 + 3], (T*)nullptr);
            }
    }
    __syncthreads();
    if (valid) {
        constexpr int CPR = ROWB / 16;
#pragma unroll
        for (int i = 0; i < 32 * CPR / 64; ++i) {
            const int c = lane + 64 * i, row = c / CPR, ch = c % CPR;
            T* orow = dqkv + (size_t)(b * L + kt * 32 + row) * ldq + h * HD + ch * TT<T>::EPC;
            *reinterpret_cast<uint4*>(orow + D) = *reinterpret_cast<const uint4*>(stg + row * ROWB + ch * 16);
            *reinterpret_cast<uint4*>(orow + 2 * D) = *reinterpret_cast<const uint4*>(stg + 32 * ROWB + row * ROWB + ch * 16);
        }
    }
}

// ------------------------------------------------------------------------------- host side
// RcatT[d][r] (T) -- the transposed operand of the dQ bias contraction
template <typename T> __global__ void relpos_pack_t_kernel(const float* rh, int nh, const float* rw, int nw, T* out, int NRP, int hd) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= NRP * hd) return;
    const int d = i / NRP, r = i % NRP;
    float v = 0.f;
    if (r < nh) v = rh[r * hd + d];
    else if (r - nh < nw) v = rw[(r - nh) * hd + d];
    out[i] = from_f<T>(v);
}
extern "C" int pa_relpos_rows_padded(int Hp, int Wp);
extern "C" int pa_relpos_pack_t(int dtype, const float* rel_pos_h, const float* rel_pos_w, void* rcatT, int Hp, int Wp, int head_dim,
                                hipStream_t st) {
    if (head_dim <= 0 || head_dim % 16) return (int)hipErrorInvalidValue;
    const int NRP = pa_relpos_rows_padded(Hp, Wp);
    const int n = NRP * head_dim;
    if (dtype == PA_BF16)
        PA_LAUNCH(relpos_pack_t_kernel<bf16>, dim3((n + 255) / 256), dim3(256), 0, st, rel_pos_h, 2 * Hp - 1, rel_pos_w, 2 * Wp - 1, (bf16*)rcatT, NRP, head_dim);
    else
        PA_LAUNCH(relpos_pack_t_kernel<float>, dim3((n + 255) / 256), dim3(256), 0, st, rel_pos_h, 2 * Hp - 1, rel_pos_w, 2 * Wp - 1, (float*)rcatT, NRP, head_dim);
    LAUNCH_CHECK();
}

extern "C" int64_t pa_attn_bwd_aux_bytes(int batch, int L, int heads, int Hp, int Wp) {
    const int64_t gen1 = (int64_t)batch * heads * (L / 32) * (Hp + Wp + 2) * 32 * sizeof(float);
    const int64_t gen2 = (attn2_ok(L, Hp, Wp, 64) || attn2_ok(L, Hp, Wp, 80)) ? attn2_aux_bytes(batch, L, heads, Hp, Wp) : 0;
    return gen1 > gen2 ? gen1 : gen2;
}

template <typename T, int NW, int HD>
static int attn_bwd_dq_launch(const T* qkv, int64_t ldq, const T* rcat, const T* rcatT, const T* dout, int64_t lddo, const float* lse,
                              const float* delta, T* dqkv, T* dG, float* aux, int Bn, int L, int H, int Hp, int Wp, float scale, size_t smem,
                              int ts, hipStream_t st) {
    const int NRP = pa_relpos_rows_padded(Hp, Wp);
    auto kern = attn_bwd_dq_kernel<T, NW, HD>;
    static bool done = false;
    if (!done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        done = true;
    }
    const int qtiles = L / 32;
    PA_LAUNCH(kern, dim3((qtiles + NW - 1) / NW, Bn * H), dim3(NW * 64), smem, st, qkv, (size_t)ldq, rcat, rcatT, dout,
              (size_t)lddo, lse, delta, dqkv, dG, aux, L, H, Hp, Wp, NRP, scale, ts);
    return (int)hipGetLastError();
}

template <typename T, int HD>
static int attn_bwd_t(const T* qkv, int64_t ldq, const T* rcat, const T* rcatT, const T* dout, int64_t lddo, const float* lse,
                      const float* delta, T* dqkv, T* dG, float* aux, int Bn, int L, int H, int Hp, int Wp, float scale, hipStream_t st) {
    typedef KvTile<T, HD> KV;
    const int TS = Hp + Wp;
    const int qtiles = L / 32;
    {   // kernel A: 4 waves per workgroup where the per-wave tables fit the LDS beside the staging images, otherwise 2
        int ts = 2 * 32 * TS * 4;
        const int stg = 32 * HD * (int)sizeof(T);
        if (ts < stg) ts = stg;
        ts = (ts + 15) / 16 * 16;
        const size_t stages = 2 * (size_t)(2 * KV::KB + KV::VB);
        int e = (int)hipErrorInvalidValue;
        if (stages + 4 * (size_t)ts <= 160 * 1024) {
            e = attn_bwd_dq_launch<T, 4, HD>(qkv, ldq, rcat, rcatT, dout, lddo, lse, delta, dqkv, dG, aux, Bn, L, H, Hp, Wp, scale, stages + 4 * (size_t)ts, ts, st);
        } else if constexpr (HD != 64) {      // 3 waves x 64 threads >= 2 * HD, as TrStage needs (HD = 64 keeps its round-1 instantiation only)
            if (stages + 3 * (size_t)ts <= 160 * 1024)
                e = attn_bwd_dq_launch<T, 3, HD>(qkv, ldq, rcat, rcatT, dout, lddo, lse, delta, dqkv, dG, aux, Bn, L, H, Hp, Wp, scale, stages + 3 * (size_t)ts, ts, st);
        }
        if (e) return e;
    }
    {   // kernel B
        constexpr int NW = 4;
        const int AUXB = (TS + 2) * AUX_LD * 4;
        size_t smem = 2 * (size_t)(2 * KV::KB + 2 * KV::VB + AUXB);
        const size_t stg = (size_t)NW * 2 * 32 * HD * sizeof(T);
        if (smem < stg) smem = stg;
        if (smem > 160 * 1024 || (TS + 2) * 8 > (HD == 64 ? 3 : 4) * NW * 64) return (int)hipErrorInvalidValue;
        auto kern = attn_bwd_dkv_kernel<T, NW, HD>;
        static bool done = false;
        if (!done) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return (int)e;
            done = true;
        }
        PA_LAUNCH(kern, dim3((qtiles + NW - 1) / NW, Bn * H), dim3(NW * 64), smem, st, qkv, (size_t)ldq, dout, (size_t)lddo,
                           aux, dqkv, L, H, Hp, Wp, scale);
        return (int)hipGetLastError();
    }
}

// dqkv: T [batch*L, 3*heads*hd] (same layout as qkv); dG: T [batch*L, heads*NRP]; aux: pa_attn_bwd_aux_bytes scratch
// 1 when the backward for this case reads Delta from the table tiles (28-token-wide bf16 kernels): pa_attn_bwd_prep replaces
// pa_attn_bwd_delta and pa_attn_bwd is called with delta = NULL
extern "C" int pa_attn_bwd_prep_ok(int dtype, int L, int Hp, int Wp, int head_dim) {
    return (dtype == PA_BF16 && head_dim == ATT_HD && L == Hp * Wp && attn3_ok(L, Hp, Wp)) ? 1 : 0;
}
extern "C" int pa_attn_bwd_prep(int dtype, const void* out, int64_t ldo, const void* dout, int64_t lddo, const float* lse, void* tables,
                                int batch, int L, int heads, int Hp, int Wp, int head_dim, float scale, hipStream_t st) {
    if (!pa_attn_bwd_prep_ok(dtype, L, Hp, Wp, head_dim) || tables == nullptr || ldo % 8 || lddo % 8) return (int)hipErrorInvalidValue;
    return attn3_bwd_prep((const bf16*)out, ldo, (const bf16*)dout, lddo, lse, tables, batch, L, heads, Hp, scale, st);
}
extern "C" int64_t pa_attn_bwd_relpos_partials_bytes(int dtype, int batch, int L, int heads, int Hp, int Wp, int head_dim) {
    if (dtype != PA_BF16 || head_dim != ATT_HD || L != Hp * Wp) return 0;
    return attn3_relpos_partials_bytes(batch, L, heads, Hp, Wp);
}
extern "C" int pa_attn_bwd(int dtype, const void* qkv, int64_t ldq, const void* rcat, const void* rcatT, const void* dout, int64_t lddo,
                           const float* lse, const float* delta, void* dqkv, void* dG, void* relpos_part, void* aux, void* tables,
                           const void* out, int64_t ldo, int batch, int L, int heads, int Hp, int Wp, int head_dim, float scale, hipStream_t st) {
    if (L != Hp * Wp || L % 32 || Hp % 4 || Wp % 4 || (head_dim != 64 && head_dim != 80)) return (int)hipErrorInvalidValue;
    if (dtype == PA_BF16 && head_dim == ATT_HD && tables != nullptr && attn3_ok(L, Hp, Wp)) {
        if (relpos_part != nullptr && attn3_relpos_partials_bytes(batch, L, heads, Hp, Wp) == 0) return (int)hipErrorInvalidValue;
        ++g_attn_counts[5];
        if (out != nullptr && ldo % 8) return (int)hipErrorInvalidValue;
        return attn3_bwd((const bf16*)qkv, ldq, (const bf16*)rcatT, (const bf16*)dout, lddo, lse, delta, tables, (bf16*)dqkv, (bf16*)dG,
                         (float*)relpos_part, batch, L, heads, Hp, Wp, scale, (const bf16*)out, ldo, st);
    }
    if (relpos_part != nullptr || dG == nullptr || delta == nullptr) return (int)hipErrorInvalidValue;      // only the generation-3 kernels fuse the rel-pos gradient / read Delta from the tables
    if (dtype == PA_BF16 && attn2_ok(L, Hp, Wp, head_dim)) {
        ++g_attn_counts[4];
        return attn2_bwd((const bf16*)qkv, ldq, (const bf16*)rcat, (const bf16*)rcatT, (const bf16*)dout, lddo, lse, delta, (bf16*)dqkv,
                         (bf16*)dG, aux, batch, L, heads, Hp, Wp, head_dim, scale, st);
    }
    ++g_attn_counts[3];
#define PA_ATTN_BWD(TT_, HD_) attn_bwd_t<TT_, HD_>((const TT_*)qkv, ldq, (const TT_*)rcat, (const TT_*)rcatT, (const TT_*)dout, lddo, lse, delta, \
                                                   (TT_*)dqkv, (TT_*)dG, (float*)aux, batch, L, heads, Hp, Wp, scale, st)
    if (dtype == PA_BF16) return head_dim == 80 ? PA_ATTN_BWD(bf16, 80) : PA_ATTN_BWD(bf16, 64);
    return head_dim == 80 ? PA_ATTN_BWD(float, 80) : PA_ATTN_BWD(float, 64);
#undef PA_ATTN_BWD
}

// d[rel_pos_h ; rel_pos_w ; pad][NRP, hd] (fp32) = sum over heads, samples, queries of dG[., r] * q[., d]
extern "C" int64_t pa_attn_bwd_relpos_workspace_bytes(int dtype, int batch, int L, int heads, int Hp, int Wp, int head_dim) {
    const int NRP = pa_relpos_rows_padded(Hp, Wp);
    return (int64_t)heads * 32 * NRP * head_dim * sizeof(float);
}
template <typename T>
static int relpos_grad_t(const T* dG, const T* qkv, int64_t ldq, float* drcat, float* ws, int Bn, int L, int H, int NRP, int hd, hipStream_t st) {
    const int R = Bn * L;
    OpT<T> A{dG, (size_t)H * NRP, NRP, (size_t)NRP};
    OpT<T> B{qkv, (size_t)ldq, hd, (size_t)hd};
    const int nku = (R + TT<T>::BK - 1) / TT<T>::BK;
    static const int want = [] { const char* v = getenv("PA_RELPOS_SPLITS"); return v ? atoi(v) : 16; }();      // 16: 45.7 us, 32: 52.6, 8: 67.9 (B=8)
    static const bool from_env = getenv("PA_RELPOS_SPLITS") != nullptr;
    int splits = (!from_env && g_relpos_splits > 0) ? g_relpos_splits : want;
    if (splits > 32) splits = 32;          // workspace bound (pa_attn_bwd_relpos_workspace_bytes)
    if (splits < 1) splits = 1;
    if (splits > nku) splits = nku;
    struct Epi {
        float* out; size_t slab; int M, N;
        DEVI void operator()(const f32x16 (&acc)[2][2], int ib, int jb, int lane, int z) const {
            float* o = out + (size_t)z * slab;
            foreach_acc(acc, ib, jb, lane, [&](int i, int j, float v) {
                if (i < M && j < N) o[(size_t)i * N + j] = v;
            });
        }
    };
    int e = launch_gemm<T, 2, 2>(A, B, Epi{ws, (size_t)NRP * hd, NRP, hd}, NRP, hd, R, splits, H, st);
    if (e) return e;
    return pa_slab_reduce(ws, drcat, (int64_t)NRP * hd, splits * H, (int64_t)NRP * hd, 0, st);
}
extern "C" int pa_attn_bwd_relpos_reduce(const void* relpos_part, float* drcat, void* workspace, int batch, int L, int heads, int Hp, int Wp,
                                         int head_dim, hipStream_t st) {
    if (head_dim != ATT_HD || relpos_part == nullptr || attn3_relpos_partials_bytes(batch, L, heads, Hp, Wp) == 0) return (int)hipErrorInvalidValue;
    return attn3_relpos_reduce((const float*)relpos_part, drcat, (float*)workspace, batch, L, heads, Hp, Wp, st);
}
extern "C" int pa_attn_bwd_relpos(int dtype, const void* dG, const void* qkv, int64_t ldq, float* drcat, void* workspace, int batch, int L,
                                  int heads, int Hp, int Wp, int head_dim, hipStream_t st) {
    if (head_dim <= 0 || head_dim % 16) return (int)hipErrorInvalidValue;
    const int NRP = pa_relpos_rows_padded(Hp, Wp);
    if (dtype == PA_BF16) return relpos_grad_t<bf16>((const bf16*)dG, (const bf16*)qkv, ldq, drcat, (float*)workspace, batch, L, heads, NRP, head_dim, st);
    return relpos_grad_t<float>((const float*)dG, (const float*)qkv, ldq, drcat, (float*)workspace, batch, L, heads, NRP, head_dim, st);
}
