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

template <typename T, int NW>
__global__ __launch_bounds__(NW * 64) void attn_fwd_kernel(const T* __restrict__ qkv, size_t ldq, const T* __restrict__ rcat,
                                                           T* __restrict__ out, size_t ldo, float* __restrict__ lse, int L, int H,
                                                           int Hp, int Wp, int NRP, float scale, int tab_stride) {
    constexpr int NT = NW * 64;
    constexpr int KVB = KvTile<T>::BYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 5;
    const int bh = blockIdx.y, b = bh / H, h = bh % H;
    const int D = H * ATT_HD, TS = Hp + Wp;
    const T* base = qkv + (size_t)b * L * ldq + h * ATT_HD;
    const T* kbase = base + D;
    const T* vbase = base + 2 * D;
    const int qt = blockIdx.x * NW + wave;
    const bool valid = qt * 32 < L;
    const int q = qt * 32 + (lane & 31);
    float* tab = reinterpret_cast<float*>(smem + 4 * KVB + (size_t)wave * tab_stride) + (lane & 31) * TS;

    Frag<T> qf[4];
    if (valid) {
#pragma unroll
        for (int s = 0; s < 4; ++s) load_gfrag<T>(qf[s], base + (size_t)q * ldq, s, g);
        build_bias_table<T>(tab, rcat, NRP, qf, q / Wp, q % Wp, Hp, Wp, lane);
    }

    RowStage<T, NT> ks;
    TrStage<T> vs;
    const int ntile = L / 32;
    ks.load(kbase, ldq, tid);
    vs.load(vbase, ldq, tid);
    ks.store(smem, tid);
    vs.store(smem + 2 * KVB, tid);
    __syncthreads();

    f32x16 oacc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { oacc[0][r] = 0.f; oacc[1][r] = 0.f; }
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
        const unsigned char* kt = smem + (j & 1) * KVB;
        const unsigned char* vt = smem + (2 + (j & 1)) * KVB;
        if (valid) {
            f32x16 sacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                Frag<T> kf;
                load_rowfrag<T>(kf, kt, lane & 31, s, g);
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
            for (int r = 0; r < 16; ++r) { oacc[0][r] *= alpha; oacc[1][r] *= alpha; }
            Frag<T> pf[2];
            pack_frag<T>(pf[0], p);
            pack_frag<T>(pf[1], p + 8);
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    Frag<T> vf;
                    load_trfrag<T>(vf, vt, db * 32 + (lane & 31), s, g);
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
            ks.store(smem + ((j + 1) & 1) * KVB, tid);
            vs.store(smem + (2 + ((j + 1) & 1)) * KVB, tid);
        }
        __syncthreads();
    }

    // epilogue: normalise, stage O^T through this wave's (now free) table region, store whole rows
    unsigned char* stg = smem + 4 * KVB + (size_t)wave * tab_stride;
    constexpr int ROWB = ATT_HD * sizeof(T);
    if (valid) {
        const float lt = l + lane_xor32(l);
        const float inv = 1.f / lt;
        if (g == 0) lse[(size_t)bh * L + q] = (m + __builtin_amdgcn_logf(lt)) * LN2_F;   // v_log_f32 = log2
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int d0 = db * 32 + 8 * rg + 4 * g;
                T* dst = reinterpret_cast<T*>(stg + (lane & 31) * ROWB) + d0;
                const float a = oacc[db][rg * 4] * inv, bb = oacc[db][rg * 4 + 1] * inv, c = oacc[db][rg * 4 + 2] * inv,
                            d = oacc[db][rg * 4 + 3] * inv;
                *reinterpret_cast<typename TT<T>::Vec4*>(dst) = cvt4(a, bb, c, d, (T*)nullptr);
            }
    }
    __syncthreads();
    if (valid) {
        constexpr int CPR = ROWB / 16;
#pragma unroll
        for (int i = 0; i < 32 * CPR / 64; ++i) {
            const int c = lane + 64 * i, row = c / CPR, ch = c % CPR;
            const uint4 v = *reinterpret_cast<const uint4*>(stg + row * ROWB + ch * 16);
            *reinterpret_cast<uint4*>(out + (size_t)(b * L + qt * 32 + row) * ldo + h * ATT_HD + ch * TT<T>::EPC) = v;
        }
    }
}

// Rcat[r][:] = rel_pos_h rows, then rel_pos_w rows, zero padded to NRP rows (T-typed operand for the bias MFMAs)
template <typename T> __global__ void relpos_pack_kernel(const float* rh, int nh, const float* rw, int nw, T* out, int NRP) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= NRP * ATT_HD) return;
    const int r = i / ATT_HD, d = i % ATT_HD;
    float v = 0.f;
    if (r < nh) v = rh[r * ATT_HD + d];
    else if (r - nh < nw) v = rw[(r - nh) * ATT_HD + d];
    out[i] = from_f<T>(v);
}
extern "C" int pa_relpos_rows_padded(int Hp, int Wp) { return ((2 * Hp - 1 + 2 * Wp - 1) + 31) / 32 * 32; }
extern "C" int pa_relpos_pack(int dtype, const float* rel_pos_h, const float* rel_pos_w, void* rcat, int Hp, int Wp, hipStream_t st) {
    const int NRP = pa_relpos_rows_padded(Hp, Wp);
    const int n = NRP * ATT_HD;
    if (dtype == PA_BF16)
        PA_LAUNCH(relpos_pack_kernel<bf16>, dim3((n + 255) / 256), dim3(256), 0, st, rel_pos_h, 2 * Hp - 1, rel_pos_w, 2 * Wp - 1, (bf16*)rcat, NRP);
    else
        PA_LAUNCH(relpos_pack_kernel<float>, dim3((n + 255) / 256), dim3(256), 0, st, rel_pos_h, 2 * Hp - 1, rel_pos_w, 2 * Wp - 1, (float*)rcat, NRP);
    LAUNCH_CHECK();
}

static int attn_tab_stride(int Hp, int Wp, int elem) {
    int a = 32 * (Hp + Wp) * 4, b = 32 * ATT_HD * elem;
    int s = a > b ? a : b;
    return (s + 15) / 16 * 16;
}
template <typename T, int NW>
static int attn_fwd_launch(const T* qkv, int64_t ldq, const T* rcat, T* out, int64_t ldo, float* lse, int Bn, int L, int H, int Hp,
                           int Wp, float scale, hipStream_t st) {
    const int NRP = pa_relpos_rows_padded(Hp, Wp);
    const int ts = attn_tab_stride(Hp, Wp, sizeof(T));
    const size_t smem = 4 * KvTile<T>::BYTES + (size_t)NW * ts;
    auto kern = attn_fwd_kernel<T, NW>;
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

// qkv: [B', L, 3, H, 64] T (row stride ldq = 3*H*64); rcat from pa_relpos_pack; out: [B'*L, H*64] T; lse: [B'*H, L] fp32
extern "C" int64_t pa_attn_tables_bytes(int dtype, int batch, int L, int heads, int Hp, int Wp) {
    return dtype == PA_BF16 ? attn3_table_bytes(batch, L, heads, Hp, Wp) : 0;
}
extern "C" int pa_attn_fwd(int dtype, const void* qkv, int64_t ldq, const void* rcat, void* out, int64_t ldo, float* lse, void* tables,
                           int batch, int L, int heads, int Hp, int Wp, float scale, hipStream_t st) {
    if (L != Hp * Wp || L % 32 || Hp % 4 || Wp % 4 || 32 % 4) return (int)hipErrorInvalidValue;
    const bool seven = ((L / 32) % 7 == 0);
    if (dtype == PA_BF16 && attn3_ok(L, Hp, Wp))
        return attn3_fwd((const bf16*)qkv, ldq, (const bf16*)rcat, (bf16*)out, ldo, lse, tables, batch, L, heads, Hp, Wp, scale, st);
    if (dtype == PA_BF16 && attn2_ok(L, Hp, Wp))
        return attn2_fwd((const bf16*)qkv, ldq, (const bf16*)rcat, (bf16*)out, ldo, lse, batch, L, heads, Hp, Wp, scale, st);
    if (dtype == PA_BF16) {
        if (seven) return attn_fwd_launch<bf16, 7>((const bf16*)qkv, ldq, (const bf16*)rcat, (bf16*)out, ldo, lse, batch, L, heads, Hp, Wp, scale, st);
        return attn_fwd_launch<bf16, 4>((const bf16*)qkv, ldq, (const bf16*)rcat, (bf16*)out, ldo, lse, batch, L, heads, Hp, Wp, scale, st);
    }
    if (seven) return attn_fwd_launch<float, 7>((const float*)qkv, ldq, (const float*)rcat, (float*)out, ldo, lse, batch, L, heads, Hp, Wp, scale, st);
    return attn_fwd_launch<float, 4>((const float*)qkv, ldq, (const float*)rcat, (float*)out, ldo, lse, batch, L, heads, Hp, Wp, scale, st);
}
