// Fused attention forward: S = scale*QK^T + decomposed rel-pos bias, online softmax, O = PV.
// Replaces bmm / add_decomposed_rel_pos / softmax / bmm / permutes at Painter/models_painter.py:76-86 and
// util/vitdet_utils.py:63-125.  The L x L logits never touch HBM.
//
// Work split: workgroup = NW waves x 32 query rows of one (sample, head); grid = (ceil(L/32/NW), B'*H).
// Per wave, "swapped" products so that every lane owns ONE query row end to end:
//   S^T[key][q]  = K_tile . Q^T      (A = K rows from LDS, B = Q fragments held in registers)
//   O^T[d][q]   += V^T_tile . P^T    (A = V^T from the transposed LDS image, B = P packed straight
//                                     from the S^T accumulator registers -- no cross-lane traffic)
// so row max / row sum are in-lane reductions plus one lane^32 exchange, and the O rescale is per lane.
// K/V tiles of 32 keys are staged global -> registers -> swizzled LDS, double buffered, one barrier per tile.
#include "attn_common.h"
#include "../../include/painter_hip.h"
#include "attn2.h"
#include "attn3.h"

template <typename T, int NW, int HD>
__global__ __launch_bounds__(NW * 64) void attn_fwd_kernel(const T* __restrict__ qkv, size_t ldq, const T* __restrict__ rcat,
                                                           T* __restrict__ out, size_t ldo, float* __restrict__ lse, int L, int H,
                                                           int Hp, int Wp, int NRP, float scale, int tab_stride) {
    typedef KvTile<T, HD> KV;
    constexpr int NT = NW * 64;
    constexpr int KB = KV::KB, VB = KV::VB, KS = KV::KS, DB = KV::DB;
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
    // LDS: K row images [2 stages], V^T images [2 stages], then one table / output-staging region per wave
    unsigned char* const kimg = smem;
    unsigned char* const vimg = smem + 2 * KB;
    unsigned char* const wreg = smem + 2 * KB + 2 * VB + (size_t)wave * tab_stride;
    float* tab = reinterpret_cast<float*>(wreg) + (lane & 31) * TS;
    KV::zero_pad(vimg, tid, NT);
    KV::zero_pad(vimg + VB, tid, NT);

    Frag<T> qf[KS];
    if (valid) {
#pragma unroll
        for (int s = 0; s < KS; ++s) load_gfrag<T>(qf[s], base + (size_t)q * ldq, s, g);
        build_bias_table<T, HD>(tab, rcat, NRP, qf, q / Wp, q % Wp, Hp, Wp, lane);
    }

    RowStage<T, NT, HD> ks;
    TrStage<T, HD> vs;
    const int ntile = L / 32;
    ks.load(kbase, ldq, tid);
    vs.load(vbase, ldq, tid);
    ks.store(kimg, tid);
    vs.store(vimg, tid);
    __syncthreads();

    f32x16 oacc[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[db][r] = 0.f;
    float m = -INFINITY, l = 0.f;
    const float sl = scale * LOG2E_F;
    // running (key row, key col) of the four 4-key runs this lane owns in the current tile
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
            ks.load(kbase + (size_t)(j + 1) * 32 * ldq, ldq, tid);
            vs.load(vbase + (size_t)(j + 1) * 32 * ldq, ldq, tid);
        }
        const unsigned char* kt = kimg + (j & 1) * KB;
        const unsigned char* vt = vimg + (j & 1) * VB;
        if (valid) {
            f32x16 sacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                Frag<T> kf;
                load_rowfrag<T, HD>(kf, kt, lane & 31, s, g);
                mma(sacc, kf, qf[s]);
            }
            float p[16];
            float tmax = -INFINITY;
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const float bhv = tab[kh[rg]];
                const float4 bw = *reinterpret_cast<const float4*>(tab + Hp + kw[rg]);
                p[rg * 4 + 0] = fmaf(sacc[rg * 4 + 0], sl, bhv + bw.x);
                p[rg * 4 + 1] = fmaf(sacc[rg * 4 + 1], sl, bhv + bw.y);
                p[rg * 4 + 2] = fmaf(sacc[rg * 4 + 2], sl, bhv + bw.z);
                p[rg * 4 + 3] = fmaf(sacc[rg * 4 + 3], sl, bhv + bw.w);
                tmax = fmaxf(tmax, fmaxf(fmaxf(p[rg * 4], p[rg * 4 + 1]), fmaxf(p[rg * 4 + 2], p[rg * 4 + 3])));
            }
            tmax = fmaxf(tmax, lane_xor32(tmax));
            const float mn = fmaxf(m, tmax);
            const float alpha = __builtin_amdgcn_exp2f(m - mn);
            float rs = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                p[r] = __builtin_amdgcn_exp2f(p[r] - mn);
                rs += p[r];
            }
            l = l * alpha + rs;
            m = mn;
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
            Frag<T> pf[2];
            pack_frag<T>(pf[0], p);
            pack_frag<T>(pf[1], p + 8);
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    Frag<T> vf;
                    load_trfrag<T, HD>(vf, vt, db * 32 + (lane & 31), s, g);
                    mma(oacc[db], vf, pf[s]);
                }
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                kh[rg] += dq_;
                kw[rg] += dr_;
                if (kw[rg] >= Wp) { kw[rg] -= Wp; kh[rg] += 1; }
            }
        }
        if (j + 1 < ntile) {
            ks.store(kimg + ((j + 1) & 1) * KB, tid);
            vs.store(vimg + ((j + 1) & 1) * VB, tid);
        }
        __syncthreads();
    }

    // epilogue: normalise, stage O^T through this wave's (now free) table region, store whole rows
    unsigned char* stg = wreg;
    constexpr int ROWB = HD * sizeof(T);
    if (valid) {
        const float lt = l + lane_xor32(l);
        const float inv = 1.f / lt;
        if (g == 0) lse[(size_t)bh * L + q] = (m + __builtin_amdgcn_logf(lt)) * LN2_F;   // v_log_f32 = log2
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int d0 = db * 32 + 8 * rg + 4 * g;
                if (d0 < HD) {                                     // rows d >= HD of the last block are padding
                    T* dst = reinterpret_cast<T*>(stg + (lane & 31) * ROWB) + d0;
                    const float a = oacc[db][rg * 4] * inv, bb = oacc[db][rg * 4 + 1] * inv, c = oacc[db][rg * 4 + 2] * inv,
                                d = oacc[db][rg * 4 + 3] * inv;
                    *reinterpret_cast<typename TT<T>::Vec4*>(dst) = cvt4(a, bb, c, d, (T*)nullptr);
                }
            }
    }
    __syncthreads();
    if (valid) {
        constexpr int CPR = ROWB / 16;
#pragma unroll
        for (int i = 0; i < 32 * CPR / 64; ++i) {
            const int c = lane + 64 * i, row = c / CPR, ch = c % CPR;
            const uint4 v = *reinterpret_cast<const uint4*>(stg + row * ROWB + ch * 16);
            *reinterpret_cast<uint4*>(out + (size_t)(b * L + qt * 32 + row) * ldo + h * HD + ch * TT<T>::EPC) = v;
        }
    }
}

// Rcat[r][:] = rel_pos_h rows, then rel_pos_w rows, zero padded to NRP rows (T-typed operand for the bias MFMAs)
template <typename T> __global__ void relpos_pack_kernel(const float* rh, int nh, const float* rw, int nw, T* out, int NRP, int hd) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= NRP * hd) return;
    const int r = i / hd, d = i % hd;
    float v = 0.f;
    if (r < nh) v = rh[r * hd + d];
    else if (r - nh < nw) v = rw[(r - nh) * hd + d];
    out[i] = from_f<T>(v);
}
extern "C" int pa_relpos_rows_padded(int Hp, int Wp) { return ((2 * Hp - 1 + 2 * Wp - 1) + 31) / 32 * 32; }
static bool head_dim_ok(int hd) { return hd == 64 || hd == 80; }       // instantiated head dims (attn_fwd_launch / attn_bwd_t)
extern "C" int pa_relpos_pack(int dtype, const float* rel_pos_h, const float* rel_pos_w, void* rcat, int Hp, int Wp, int head_dim,
                              hipStream_t st) {
    if (head_dim <= 0 || head_dim % 16) return (int)hipErrorInvalidValue;
    const int NRP = pa_relpos_rows_padded(Hp, Wp);
    const int n = NRP * head_dim;
    if (dtype == PA_BF16)
        PA_LAUNCH(relpos_pack_kernel<bf16>, dim3((n + 255) / 256), dim3(256), 0, st, rel_pos_h, 2 * Hp - 1, rel_pos_w, 2 * Wp - 1, (bf16*)rcat, NRP, head_dim);
    else
        PA_LAUNCH(relpos_pack_kernel<float>, dim3((n + 255) / 256), dim3(256), 0, st, rel_pos_h, 2 * Hp - 1, rel_pos_w, 2 * Wp - 1, (float*)rcat, NRP, head_dim);
    LAUNCH_CHECK();
}

// Rcat and Rcat^T of EVERY block in one launch (round 6): blockIdx.y = block, blockIdx.z = 0 -> rcat [nblocks][NRP][hd], 1 -> rcatT [nblocks][hd][NRP].
// `tabs`: device array of 2 * nblocks pointers, rel_pos_h of every block, then rel_pos_w of every block.  Same values as the two per-block
// kernels (pa_relpos_pack / pa_relpos_pack_t), bit for bit; a training step re-packs all blocks with one launch instead of 2 per block.
template <typename T> __global__ void relpos_pack_batch_kernel(const float* const* __restrict__ tabs, int nblocks, int nh, int nw, T* __restrict__ rcat,
                                                               T* __restrict__ rcatT, int NRP, int hd) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= NRP * hd) return;
    const int blk = blockIdx.y;
    const bool tr = blockIdx.z != 0;
    const int r = tr ? i % NRP : i / hd, d = tr ? i / NRP : i % hd;
    const float* rh = tabs[blk];
    const float* rw = tabs[nblocks + blk];
    float v = 0.f;
    if (r < nh) v = rh[r * hd + d];
    else if (r - nh < nw) v = rw[(r - nh) * hd + d];
    (tr ? rcatT : rcat)[(size_t)blk * NRP * hd + i] = from_f<T>(v);
}
extern "C" int pa_relpos_pack_batch(int dtype, const void* tabs, void* rcat, void* rcatT, int nblocks, int Hp, int Wp, int head_dim, hipStream_t st) {
    if (head_dim <= 0 || head_dim % 16 || nblocks <= 0 || nblocks > 65535) return (int)hipErrorInvalidValue;
    const int NRP = pa_relpos_rows_padded(Hp, Wp);
    const int n = NRP * head_dim;
    const float* const* t = reinterpret_cast<const float* const*>(tabs);
    if (dtype == PA_BF16)
        PA_LAUNCH(relpos_pack_batch_kernel<bf16>, dim3((n + 255) / 256, nblocks, 2), dim3(256), 0, st, t, nblocks, 2 * Hp - 1, 2 * Wp - 1, (bf16*)rcat, (bf16*)rcatT, NRP, head_dim);
    else
        PA_LAUNCH(relpos_pack_batch_kernel<float>, dim3((n + 255) / 256, nblocks, 2), dim3(256), 0, st, t, nblocks, 2 * Hp - 1, 2 * Wp - 1, (float*)rcat, (float*)rcatT, NRP, head_dim);
    LAUNCH_CHECK();
}

static int attn_tab_stride(int Hp, int Wp, int hd, int elem) {
    int a = 32 * (Hp + Wp) * 4, b = 32 * hd * elem;
    int s = a > b ? a : b;
    return (s + 15) / 16 * 16;
}
template <typename T, int NW, int HD>
static int attn_fwd_launch(const T* qkv, int64_t ldq, const T* rcat, T* out, int64_t ldo, float* lse, int Bn, int L, int H, int Hp,
                           int Wp, float scale, hipStream_t st) {
    const int NRP = pa_relpos_rows_padded(Hp, Wp);
    const int ts = attn_tab_stride(Hp, Wp, HD, sizeof(T));
    const size_t smem = 2 * KvTile<T, HD>::KB + 2 * KvTile<T, HD>::VB + (size_t)NW * ts;
    auto kern = attn_fwd_kernel<T, NW, HD>;
    static bool done = false;
    if (!done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        done = true;
    }
    if (smem > 160 * 1024) return (int)hipErrorInvalidValue;
    const int qtiles = L / 32;
    dim3 grid((qtiles + NW - 1) / NW, Bn * H);
    PA_LAUNCH(kern, grid, dim3(NW * 64), smem, st, qkv, (size_t)ldq, rcat, out, (size_t)ldo, lse, L, H, Hp, Wp, NRP, scale, ts);
    return (int)hipGetLastError();
}
template <typename T>
static int attn_fwd_t(const void* qkv, int64_t ldq, const void* rcat, void* out, int64_t ldo, float* lse, int Bn, int L, int H, int Hp,
                      int Wp, int hd, float scale, hipStream_t st) {
    const T *q = (const T*)qkv, *r = (const T*)rcat;
    T* o = (T*)out;
    if (hd == 80) return attn_fwd_launch<T, 4, 80>(q, ldq, r, o, ldo, lse, Bn, L, H, Hp, Wp, scale, st);
    if ((L / 32) % 7 == 0) return attn_fwd_launch<T, 7, 64>(q, ldq, r, o, ldo, lse, Bn, L, H, Hp, Wp, scale, st);
    return attn_fwd_launch<T, 4, 64>(q, ldq, r, o, ldo, lse, Bn, L, H, Hp, Wp, scale, st);
}

// qkv: [B', L, 3, H, hd] T (row stride ldq = 3*H*hd); rcat from pa_relpos_pack; out: [B'*L, H*hd] T; lse: [B'*H, L] fp32
extern "C" int64_t pa_attn_tables_bytes(int dtype, int batch, int L, int heads, int Hp, int Wp, int head_dim) {
    return dtype == PA_BF16 && head_dim == ATT_HD ? attn3_table_bytes(batch, L, heads, Hp, Wp) : 0;
}
extern "C" int pa_attn_fwd(int dtype, const void* qkv, int64_t ldq, const void* rcat, void* out, int64_t ldo, float* lse, void* tables,
                           int batch, int L, int heads, int Hp, int Wp, int head_dim, float scale, hipStream_t st) {
    if (L != Hp * Wp || L % 32 || Hp % 4 || Wp % 4 || !head_dim_ok(head_dim)) return (int)hipErrorInvalidValue;
    if (dtype == PA_BF16 && head_dim == ATT_HD && attn3_ok(L, Hp, Wp)) {
        ++g_attn_counts[2];
        return attn3_fwd((const bf16*)qkv, ldq, (const bf16*)rcat, (bf16*)out, ldo, lse, tables, batch, L, heads, Hp, Wp, scale, st);
    }
    if (dtype == PA_BF16 && attn2_ok(L, Hp, Wp, head_dim)) {
        ++g_attn_counts[1];
        return attn2_fwd((const bf16*)qkv, ldq, (const bf16*)rcat, (bf16*)out, ldo, lse, batch, L, heads, Hp, Wp, head_dim, scale, st);
    }
    ++g_attn_counts[0];
    if (dtype == PA_BF16) return attn_fwd_t<bf16>(qkv, ldq, rcat, out, ldo, lse, batch, L, heads, Hp, Wp, head_dim, scale, st);
    return attn_fwd_t<float>(qkv, ldq, rcat, out, ldo, lse, batch, L, heads, Hp, Wp, head_dim, scale, st);
}

extern "C" int pa_attn_launch_counts(long long* out6) {
    if (out6 == nullptr) return (int)hipErrorInvalidValue;
    for (int i = 0; i < 6; ++i) out6[i] = g_attn_counts[i];
    return 0;
}
