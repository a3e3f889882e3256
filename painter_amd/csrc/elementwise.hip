// HBM-bound pieces of the path: abs-pos resize operator, token-assembly backward, stream merge, SegGPT feature
// ensemble, masked smooth-L1 loss (+ ignore rule) forward/backward, patchify, decoder-tail point-wise backward,
// parameter casts.  Reference lines are cited per kernel (paths relative to the reference root).
#include "common.h"
#include "../../include/painter_hip.h"

extern "C" int pa_slab_reduce(const float* in, float* out, int64_t n, int nz, int64_t stride, int accumulate, hipStream_t st);

// ------------------------------------------------------------------------------- casts (autocast's weight casts, once per step)
__global__ void cast_bf16_kernel(const float* __restrict__ in, bf16* __restrict__ out, size_t n) {
    size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 4 <= n) {
        const float4 v = *reinterpret_cast<const float4*>(in + i);
        *reinterpret_cast<uint2*>(out + i) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
    } else {
        for (; i < n; ++i) out[i] = (bf16)in[i];
    }
}
extern "C" int pa_cast_bf16(const float* in, void* out, int64_t n, hipStream_t st) {
    if (n <= 0) return 0;
    PA_LAUNCH(cast_bf16_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, st, in, (bf16*)out, (size_t)n);
    LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------- abs pos: pos = M . pos_embed[0, skip:]
// get_abs_pos (util/vitdet_utils.py:128-157) as the constant bicubic operator M [L, S], applied in its
// row-sparse form (hostmath.sparse_rows): out[r][n] = sum_j val[r][j] * (x0[idx[r][j]][n] + x1[idx[r][j]][n]), K entries per row
// (padding entries have val 0).  Forward: rows = tokens, 16 entries each; backward (M^T): rows = source cells.  The dense form cost
// 67 / 266 us per step (a dependent load of M per (token, source) pair); this one streams K short rows.
// block = 64 columns x 4 entry-lanes; partial sums are combined through LDS in a fixed order (deterministic)
__global__ __launch_bounds__(256) void sparse_rows_kernel(const int* __restrict__ idx, const float* __restrict__ val, int K, const float* __restrict__ x0,
                                                          const float* __restrict__ x1, float* __restrict__ out, int D) {
    __shared__ float red[4][64];
    const int r = blockIdx.y, cl = threadIdx.x & 63, jl = threadIdx.x >> 6, n = blockIdx.x * 64 + cl;
    float acc = 0.f;
    if (n < D) {
        for (int j = jl; j < K; j += 4) {
            const float m = val[(size_t)r * K + j];
            if (m == 0.f) continue;            // padding entries (idx 0, val 0) must not touch row 0: 0 * inf would spread a NaN of an overflowed step
            const size_t o = (size_t)idx[(size_t)r * K + j] * D + n;
            acc = fmaf(m, x1 ? x0[o] + x1[o] : x0[o], acc);
        }
    }
    red[jl][cl] = acc;
    __syncthreads();
    if (jl == 0 && n < D) out[(size_t)r * D + n] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}
// pe / dpe point at the first non-cls row of pos_embed ([S, D]); (idx, val, K) = sparse_rows(M) for the forward, sparse_rows(M^T) for the backward
extern "C" int pa_pos_fwd(const int* idx, const float* val, int K, const float* pe, float* pos, int L, int D, hipStream_t st) {
    if (K < 1) return (int)hipErrorInvalidValue;
    PA_LAUNCH(sparse_rows_kernel, dim3((D + 63) / 64, L), dim3(256), 0, st, idx, val, K, pe, (const float*)nullptr, pos, D);
    LAUNCH_CHECK();
}
extern "C" int pa_pos_bwd(const int* idxT, const float* valT, int KT, const float* gx, const float* gy, float* dpe, int S, int D, hipStream_t st) {
    if (KT < 1) return (int)hipErrorInvalidValue;
    PA_LAUNCH(sparse_rows_kernel, dim3((D + 63) / 64, S), dim3(256), 0, st, idxT, valT, KT, gx, gy, dpe, D);
    LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------- token assembly backward (models_painter.py:392-409)
// dPE[x rows] = dx; dPE[y rows] = dy * (1 - w); sums[0][l] = sum_b dx, sums[1][l] = sum_b dy, sums[2][l] = sum_b w dy
template <typename T>
__global__ void tokens_bwd_kernel(const float* __restrict__ dx0, const unsigned char* __restrict__ mask, int mbs, T* __restrict__ dpe,
                                  float* __restrict__ sums, int Bn, int L, int D) {
    const int l = blockIdx.y, n = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (n >= D) return;
    float4 sx = make_float4(0, 0, 0, 0), sy = sx, sm = sx;
    for (int b = 0; b < Bn; ++b) {
        const size_t rx = ((size_t)b * L + l) * D + n, ry = ((size_t)(Bn + b) * L + l) * D + n;
        const float4 gx = *reinterpret_cast<const float4*>(dx0 + rx), gy = *reinterpret_cast<const float4*>(dx0 + ry);
        const float w = mask[(size_t)b * mbs + l] ? 1.f : 0.f;
        sx.x += gx.x; sx.y += gx.y; sx.z += gx.z; sx.w += gx.w;
        sy.x += gy.x; sy.y += gy.y; sy.z += gy.z; sy.w += gy.w;
        sm.x += w * gy.x; sm.y += w * gy.y; sm.z += w * gy.z; sm.w += w * gy.w;
        const float k = 1.f - w;
        *reinterpret_cast<typename TT<T>::Vec4*>(dpe + rx) = cvt4(gx.x, gx.y, gx.z, gx.w, (T*)nullptr);
        *reinterpret_cast<typename TT<T>::Vec4*>(dpe + ry) = cvt4(gy.x * k, gy.y * k, gy.z * k, gy.w * k, (T*)nullptr);
    }
    const size_t o = (size_t)l * D + n, LD = (size_t)L * D;
    *reinterpret_cast<float4*>(sums + o) = sx;
    *reinterpret_cast<float4*>(sums + LD + o) = sy;
    *reinterpret_cast<float4*>(sums + 2 * LD + o) = sm;
}
extern "C" int pa_tokens_bwd(int dtype, const float* dx0, const unsigned char* mask, int mask_batch_stride, void* dpe, float* sums,
                             int batch, int L, int D, hipStream_t st) {
    if (D % 4) return (int)hipErrorInvalidValue;
    dim3 grid((D / 4 + 63) / 64, L);
    if (dtype == PA_BF16) PA_LAUNCH(tokens_bwd_kernel<bf16>, grid, dim3(64), 0, st, dx0, mask, mask_batch_stride, (bf16*)dpe, sums, batch, L, D);
    else PA_LAUNCH(tokens_bwd_kernel<float>, grid, dim3(64), 0, st, dx0, mask, mask_batch_stride, (float*)dpe, sums, batch, L, D);
    LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------- stream merge (models_painter.py:414-415)
__global__ void merge_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, size_t n) {
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    const float4 a = *reinterpret_cast<const float4*>(x + i), b = *reinterpret_cast<const float4*>(x + n + i);
    *reinterpret_cast<float4*>(out + i) = make_float4((a.x + b.x) * 0.5f, (a.y + b.y) * 0.5f, (a.z + b.z) * 0.5f, (a.w + b.w) * 0.5f);
}
extern "C" int pa_merge_fwd(const float* x, float* out, int64_t n_out, hipStream_t st) {
    if (n_out % 4) return (int)hipErrorInvalidValue;
    PA_LAUNCH(merge_fwd_kernel, dim3((unsigned)((n_out / 4 + 255) / 256)), dim3(256), 0, st, x, out, (size_t)n_out);
    LAUNCH_CHECK();
}
// dx[both halves] = 0.5 * dmerged; dxT = rowscale[row / rps] * dx (T copy for the next GEMM operand)
template <typename T>
__global__ void merge_bwd_kernel(const float* __restrict__ dm, float* __restrict__ dx, T* __restrict__ dxT, const float* __restrict__ rowscale,
                                 int rps, size_t rows_half, int D) {
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const size_t n = rows_half * D;
    if (i >= n) return;
    const float4 g = *reinterpret_cast<const float4*>(dm + i);
    const float4 h = make_float4(g.x * 0.5f, g.y * 0.5f, g.z * 0.5f, g.w * 0.5f);
    *reinterpret_cast<float4*>(dx + i) = h;
    *reinterpret_cast<float4*>(dx + n + i) = h;
    const size_t row = i / D;
    const float s0 = rowscale ? rowscale[row / rps] : 1.f, s1 = rowscale ? rowscale[(row + rows_half) / rps] : 1.f;
    *reinterpret_cast<typename TT<T>::Vec4*>(dxT + i) = cvt4(h.x * s0, h.y * s0, h.z * s0, h.w * s0, (T*)nullptr);
    *reinterpret_cast<typename TT<T>::Vec4*>(dxT + n + i) = cvt4(h.x * s1, h.y * s1, h.z * s1, h.w * s1, (T*)nullptr);
}
extern "C" int pa_merge_bwd(int dtype, const float* dmerged, float* dx, void* dxT, const float* rowscale, int rows_per_sample,
                            int64_t rows_half, int D, hipStream_t st) {
    if (D % 4) return (int)hipErrorInvalidValue;
    const size_t n = (size_t)rows_half * D;
    dim3 grid((unsigned)((n / 4 + 255) / 256));
    if (dtype == PA_BF16) PA_LAUNCH(merge_bwd_kernel<bf16>, grid, dim3(256), 0, st, dmerged, dx, (bf16*)dxT, rowscale, rows_per_sample, (size_t)rows_half, D);
    else PA_LAUNCH(merge_bwd_kernel<float>, grid, dim3(256), 0, st, dmerged, dx, (float*)dxT, rowscale, rows_per_sample, (size_t)rows_half, D);
    LAUNCH_CHECK();
}
// scaled T copy of an fp32 matrix: out = rowscale[row / rps] * in
template <typename T>
__global__ void scale_cast_kernel(const float* __restrict__ in, T* __restrict__ out, const float* __restrict__ rowscale, int rps, size_t n, int D) {
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    const float4 g = *reinterpret_cast<const float4*>(in + i);
    const float s = rowscale ? rowscale[(i / D) / rps] : 1.f;
    *reinterpret_cast<typename TT<T>::Vec4*>(out + i) = cvt4(g.x * s, g.y * s, g.z * s, g.w * s, (T*)nullptr);
}
extern "C" int pa_scale_cast(int dtype, const float* in, void* out, const float* rowscale, int rows_per_sample, int64_t rows, int D,
                             hipStream_t st) {
    if (D % 4) return (int)hipErrorInvalidValue;
    const size_t n = (size_t)rows * D;
    dim3 grid((unsigned)((n / 4 + 255) / 256));
    if (dtype == PA_BF16) PA_LAUNCH(scale_cast_kernel<bf16>, grid, dim3(256), 0, st, in, (bf16*)out, rowscale, rows_per_sample, n, D);
    else PA_LAUNCH(scale_cast_kernel<float>, grid, dim3(256), 0, st, in, (float*)out, rowscale, rows_per_sample, n, D);
    LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------- SegGPT feature ensemble (models_seggpt.py:220-232)
// x1 = x0 + ens(a): tokens of the query half (l >= L/2) use the mean of `a` over the samples of their group.
__global__ void ensemble_resid_kernel(const float* __restrict__ x0, const float* __restrict__ a, float* __restrict__ x1, int Bn, int G,
                                      int L, int D) {
    const int l = blockIdx.y, n = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (n >= D) return;
    const bool ens = l >= L / 2;
    for (int g0 = 0; g0 < Bn; g0 += G) {
        float4 m = make_float4(0, 0, 0, 0);
        if (ens) {
            for (int b = g0; b < g0 + G; ++b) {
                const float4 v = *reinterpret_cast<const float4*>(a + ((size_t)b * L + l) * D + n);
                m.x += v.x; m.y += v.y; m.z += v.z; m.w += v.w;
            }
            const float inv = 1.f / (float)G;
            m.x *= inv; m.y *= inv; m.z *= inv; m.w *= inv;
        }
        for (int b = g0; b < g0 + G; ++b) {
            const size_t o = ((size_t)b * L + l) * D + n;
            const float4 r = *reinterpret_cast<const float4*>(x0 + o);
            const float4 v = ens ? m : *reinterpret_cast<const float4*>(a + o);
            *reinterpret_cast<float4*>(x1 + o) = make_float4(r.x + v.x, r.y + v.y, r.z + v.z, r.w + v.w);
        }
    }
}
extern "C" int pa_ensemble_resid(const float* x0, const float* a, float* x1, int batch, int group, int L, int D, hipStream_t st) {
    if (D % 4 || group <= 0 || batch % group) return (int)hipErrorInvalidValue;
    PA_LAUNCH(ensemble_resid_kernel, dim3((D / 4 + 63) / 64, L), dim3(64), 0, st, x0, a, x1, batch, group, L, D);
    LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------- masked loss (models_painter.py:433-462)
__constant__ float c_mean[3] = {0.485f, 0.456f, 0.406f};
__constant__ float c_std[3] = {0.229f, 0.224f, 0.225f};
#define LOSS_BLK 256
#define LOSS_CHUNK 4096      // elements per block

DEVI float block_sum(float v, float* sh) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) sh[w] = v;
    __syncthreads();
    float r = 0.f;
    if (threadIdx.x == 0) for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += sh[i];
    __syncthreads();
    return r;      // valid on thread 0
}
DEVI int pix_mask(const unsigned char* mask, int mbs, int b, int rem, int Wi, int P, int Wp) {
    const int y = rem / Wi, x = rem % Wi;
    return mask[(size_t)b * mbs + (y / P) * Wp + x / P] ? 1 : 0;
}
// stage 1 of the ignore rule (:444-448): partial sums of the de-normalised unmasked target
__global__ void loss_ignore_part_kernel(const float* __restrict__ tgts, const unsigned char* __restrict__ mask, int mbs, float* __restrict__ part,
                                        int HW, int Wi, int P, int Wp) {
    __shared__ float sh[LOSS_BLK / 64];
    const int b = blockIdx.y, n = 3 * HW;
    float s = 0.f;
    const int e0 = blockIdx.x * LOSS_CHUNK;
    for (int e = e0 + threadIdx.x; e < min(n, e0 + LOSS_CHUNK); e += LOSS_BLK) {
        const int c = e / HW, rem = e % HW;
        const int m = pix_mask(mask, mbs, b, rem, Wi, P, Wp);
        s += (tgts[(size_t)b * n + e] * c_std[c] + c_mean[c]) * (1.f - (float)m);
    }
    const float r = block_sum(s, sh);
    if (threadIdx.x == 0) part[(size_t)b * gridDim.x + blockIdx.x] = r;
}
__global__ void loss_ignore_flag_kernel(const float* __restrict__ part, int nblk, float* __restrict__ flag, float thresh) {
    __shared__ float sh[LOSS_BLK / 64];
    const int b = blockIdx.x;
    float s = 0.f;
    for (int i = threadIdx.x; i < nblk; i += LOSS_BLK) s += part[(size_t)b * nblk + i];
    const float r = block_sum(s, sh);
    if (threadIdx.x == 0) flag[b] = (r < thresh) ? 1.f : 0.f;
}
DEVI float loss_elem(float d, int kind, float beta) {
    const float a = fabsf(d);
    if (kind == 0) return a < beta ? 0.5f * d * d / beta : a - 0.5f * beta;     // smooth_l1(beta)
    if (kind == 1) return a;
    if (kind == 2) return d * d;
    return (a + d * d) * 0.5f;
}
DEVI float loss_grad_elem(float d, int kind, float beta) {
    const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    if (kind == 0) return fabsf(d) < beta ? d / beta : sg;
    if (kind == 1) return sg;
    if (kind == 2) return 2.f * d;
    return (sg + 2.f * d) * 0.5f;
}
__global__ void loss_part_kernel(const float* __restrict__ pred, const float* __restrict__ tgts, float* valid, const unsigned char* __restrict__ mask,
                                 int mbs, const float* __restrict__ flag, float* __restrict__ part, int HW, int Wi, int P, int Wp, int kind,
                                 float beta) {
    __shared__ float sh[LOSS_BLK / 64];
    const int b = blockIdx.y, n = 3 * HW;
    const bool ign = flag && flag[b] != 0.f;
    float num = 0.f, den = 0.f;
    const int e0 = blockIdx.x * LOSS_CHUNK;
    for (int e = e0 + threadIdx.x; e < min(n, e0 + LOSS_CHUNK); e += LOSS_BLK) {
        const size_t idx = (size_t)b * n + e;
        float v = valid[idx];
        if (ign) { v = 0.f; valid[idx] = 0.f; }           // in-place, like the reference (:448)
        const float mv = (float)pix_mask(mask, mbs, b, e % HW, Wi, P, Wp) * v;
        num += loss_elem(pred[idx] - tgts[idx], kind, beta) * mv;
        den += mv;
    }
    const float rn = block_sum(num, sh);
    const float rd = block_sum(den, sh);
    if (threadIdx.x == 0) {
        const size_t o = ((size_t)b * gridDim.x + blockIdx.x) * 2;
        part[o] = rn;
        part[o + 1] = rd;
    }
}
__global__ void loss_final_kernel(const float* __restrict__ part, int nparts, float eps_den, float* __restrict__ out) {
    __shared__ float sh[LOSS_BLK / 64];
    float num = 0.f, den = 0.f;
    for (int i = threadIdx.x; i < nparts; i += LOSS_BLK) { num += part[2 * (size_t)i]; den += part[2 * (size_t)i + 1]; }
    const float rn = block_sum(num, sh);
    const float rd = block_sum(den, sh);
    if (threadIdx.x == 0) {
        out[0] = rn / (rd + eps_den);      // loss
        out[1] = rd + eps_den;             // denominator (for backward)
    }
}
extern "C" int64_t pa_loss_workspace_bytes(int batch, int Hi, int Wi) {
    const int nblk = (3 * Hi * Wi + LOSS_CHUNK - 1) / LOSS_CHUNK;
    return (int64_t)batch * nblk * 3 * sizeof(float) + (int64_t)batch * sizeof(float) + 64;
}
// out: f32[2] = {loss, denominator}.  ignore_rule=1 (Painter): samples with unmasked target sum < 300 get valid := 0 IN PLACE.
extern "C" int pa_loss_fwd(const float* pred, const float* tgts, float* valid, const unsigned char* mask, int mask_batch_stride,
                           float* out, void* workspace, int batch, int Hi, int Wi, int P, int ignore_rule, float eps_den, int kind,
                           float beta, hipStream_t st) {
    const int HW = Hi * Wi, Wp = Wi / P;
    const int nblk = (3 * HW + LOSS_CHUNK - 1) / LOSS_CHUNK;
    float* part = reinterpret_cast<float*>(workspace);                 // [batch][nblk][2]
    float* ipart = part + (size_t)batch * nblk * 2;                    // [batch][nblk]
    float* flag = ipart + (size_t)batch * nblk;                        // [batch]
    if (ignore_rule) {
        PA_LAUNCH(loss_ignore_part_kernel, dim3(nblk, batch), dim3(LOSS_BLK), 0, st, tgts, mask, mask_batch_stride, ipart, HW, Wi, P, Wp);
        PA_LAUNCH(loss_ignore_flag_kernel, dim3(batch), dim3(LOSS_BLK), 0, st, ipart, nblk, flag, 300.f);
    }
    PA_LAUNCH(loss_part_kernel, dim3(nblk, batch), dim3(LOSS_BLK), 0, st, pred, tgts, valid, mask, mask_batch_stride,
                       ignore_rule ? flag : (const float*)nullptr, part, HW, Wi, P, Wp, kind, beta);
    PA_LAUNCH(loss_final_kernel, dim3(1), dim3(LOSS_BLK), 0, st, part, batch * nblk, eps_den, out);
    LAUNCH_CHECK();
}
__global__ void loss_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ tgts, const float* __restrict__ valid,
                                const unsigned char* __restrict__ mask, int mbs, const float* __restrict__ dloss, const float* __restrict__ lossden,
                                float* __restrict__ dpred, int HW, int Wi, int P, int Wp, int kind, float beta) {
    const int b = blockIdx.y, n = 3 * HW;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const size_t idx = (size_t)b * n + e;
    const float mv = (float)pix_mask(mask, mbs, b, e % HW, Wi, P, Wp) * valid[idx];
    dpred[idx] = dloss[0] / lossden[1] * mv * loss_grad_elem(pred[idx] - tgts[idx], kind, beta);
}
extern "C" int pa_loss_bwd(const float* pred, const float* tgts, const float* valid, const unsigned char* mask, int mask_batch_stride,
                           const float* dloss, const float* loss_out, float* dpred, int batch, int Hi, int Wi, int P, int kind,
                           float beta, hipStream_t st) {
    const int HW = Hi * Wi;
    PA_LAUNCH(loss_bwd_kernel, dim3((3 * HW + 255) / 256, batch), dim3(256), 0, st, pred, tgts, valid, mask, mask_batch_stride,
                       dloss, loss_out, dpred, HW, Wi, P, Wi / P, kind, beta);
    LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------- patchify (models_painter.py:355-368), pure index math
__global__ void patchify_kernel(const float* __restrict__ img, float* __restrict__ out, int Hp, int Wp, int P, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int PP3 = P * P * 3, L = Hp * Wp;
    const int k = (int)(i % PP3);
    const size_t t = i / PP3;
    const int l = (int)(t % L), b = (int)(t / L);
    const int c = k % 3, q = (k / 3) % P, p = k / (3 * P), h = l / Wp, w = l % Wp;
    out[i] = img[(((size_t)b * 3 + c) * Hp * P + h * P + p) * (size_t)(Wp * P) + w * P + q];
}
extern "C" int pa_patchify(const float* img, float* out, int batch, int Hp, int Wp, int P, hipStream_t st) {
    const size_t n = (size_t)batch * Hp * Wp * P * P * 3;
    PA_LAUNCH(patchify_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, img, out, Hp, Wp, P, n);
    LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------- decoder tail point-wise backward
// Through Conv1x1(64->3), GELU, LayerNorm2D(64) (models_painter.py:328-333, util/vitdet_utils.py:204-209): per pixel
// 16 lanes x 4 channels; parameter gradients accumulate in registers and leave as one partial row per workgroup:
// part[blk][0:64]=dgamma [64:128]=dbeta [128:320]=dW1[3][64] [320:323]=db1.
#define TAILP 324
// Round 5 rewrite (was 16 lanes x 4 channels per pixel, scalar fp32, an integer division per pixel: 434 us, VALU busy 1.0, for 0.86 GB of
// traffic): 8 lanes x 8 channels per pixel = one 16-byte load / store per lane, the channel reductions take three DPP steps for eight
// values instead of four for four, the element-wise chains run on float2 values (v_pk_fma / v_pk_mul_f32: half the issue slots; packed
// fp32 is exact, common.h), GELU and GELU' share their transcendentals (gelu_parts2), and (sample, pixel) advance incrementally.
// A wave = 8 pixels per iteration; the parameter-gradient accumulators are combined across the wave's 8 pixel slots with DPP / permlane
// steps at the end (fixed order), so LDS only holds one partial row per wave.
DEVI float oct_sum(float v) { v = quad_sum(v); v += lane_dpp<0x141>(v); return v; }          // over the 8 lanes of a pixel; every lane gets it
template <typename T>
__global__ __launch_bounds__(256) void tail_bwd_kernel(const float* __restrict__ dpred, const T* __restrict__ y3, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ w1, T* __restrict__ dy3,
                                                       float* __restrict__ part, int HW, int npix, float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane & 7, slot = (threadIdx.x >> 3);             // 32 pixel slots per workgroup
    const int c0 = sub * 8;
    f32x2_t g2[4], b2[4], w2[3][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        g2[k] = (f32x2_t){gamma[c0 + 2 * k], gamma[c0 + 2 * k + 1]};
        b2[k] = (f32x2_t){beta[c0 + 2 * k], beta[c0 + 2 * k + 1]};
#pragma unroll
        for (int o = 0; o < 3; ++o) w2[o][k] = (f32x2_t){w1[o * 64 + c0 + 2 * k], w1[o * 64 + c0 + 2 * k + 1]};
    }
    const f32x2_t zero2 = {0.f, 0.f};
    f32x2_t ag[4] = {zero2, zero2, zero2, zero2}, ab[4] = {zero2, zero2, zero2, zero2};
    f32x2_t aw[3][4] = {{zero2, zero2, zero2, zero2}, {zero2, zero2, zero2, zero2}, {zero2, zero2, zero2, zero2}};
    float ab1[3] = {0.f, 0.f, 0.f};
    const int stride = gridDim.x * 32;
    // Inputs travel TWO iterations ahead of their use (a ring of two register sets): with ~150 registers only 12 waves fit a CU, and one
    // 1 KB request per wave in flight would cap the chip at ~2 TB/s; two more per wave keep ~30 KB per CU outstanding.
    struct In { uint4 a, b; float d0, d1, d2; };
    struct Pos { int pix, b, rem; };
    auto advance = [&](Pos& p) {
        p.pix += stride;
        p.rem += stride;
        while (p.rem >= HW) { p.rem -= HW; ++p.b; }
    };
    auto fetch = [&](const Pos& p) {
        In in;
        const int q = min(p.pix, npix - 1);                             // past the end: a valid address, the value is never used
        const int bb = p.pix < npix ? p.b : (npix - 1) / HW, rr = p.pix < npix ? p.rem : (npix - 1) % HW;
        in.a = *reinterpret_cast<const uint4*>(y3 + (size_t)q * 64 + c0);
        if constexpr (sizeof(T) == 4) in.b = *reinterpret_cast<const uint4*>(y3 + (size_t)q * 64 + c0 + 4);
        else in.b = in.a;
        const float* dp = dpred + (size_t)bb * 3 * HW + rr;
        in.d0 = dp[0]; in.d1 = dp[(size_t)HW]; in.d2 = dp[(size_t)2 * HW];
        return in;
    };
    Pos cur, nxt;
    cur.pix = blockIdx.x * 32 + slot;
    cur.b = cur.pix / HW;
    cur.rem = cur.pix - cur.b * HW;
    nxt = cur;
    In in0 = fetch(nxt);
    advance(nxt);
    In in1 = fetch(nxt);
    advance(nxt);
    for (; cur.pix < npix; advance(cur)) {
        const In in = in0;
        in0 = in1;
        in1 = fetch(nxt);
        advance(nxt);
        const int pix = cur.pix;
        f32x2_t y[4];
        if constexpr (sizeof(T) == 2) {
            const uint4 raw = in.a;
            y[0] = (f32x2_t){bf16_lo(raw.x), bf16_hi(raw.x)}; y[1] = (f32x2_t){bf16_lo(raw.y), bf16_hi(raw.y)};
            y[2] = (f32x2_t){bf16_lo(raw.z), bf16_hi(raw.z)}; y[3] = (f32x2_t){bf16_lo(raw.w), bf16_hi(raw.w)};
        } else {
            y[0] = (f32x2_t){__builtin_bit_cast(float, in.a.x), __builtin_bit_cast(float, in.a.y)};
            y[1] = (f32x2_t){__builtin_bit_cast(float, in.a.z), __builtin_bit_cast(float, in.a.w)};
            y[2] = (f32x2_t){__builtin_bit_cast(float, in.b.x), __builtin_bit_cast(float, in.b.y)};
            y[3] = (f32x2_t){__builtin_bit_cast(float, in.b.z), __builtin_bit_cast(float, in.b.w)};
        }
        const float d0 = in.d0, d1 = in.d1, d2 = in.d2;
        f32x2_t s2 = (y[0] + y[1]) + (y[2] + y[3]);
        const float mu = oct_sum(s2[0] + s2[1]) * (1.f / 64);
        const f32x2_t mu2 = {mu, mu};
        f32x2_t q2 = zero2;
#pragma unroll
        for (int k = 0; k < 4; ++k) { y[k] = y[k] - mu2; q2 = __builtin_elementwise_fma(y[k], y[k], q2); }
        const float rs = 1.f / sqrtf(oct_sum(q2[0] + q2[1]) * (1.f / 64) + eps);
        const f32x2_t rs2 = {rs, rs}, dd0 = {d0, d0}, dd1 = {d1, d1}, dd2 = {d2, d2};
        f32x2_t dxh[4], m1 = zero2, m2 = zero2;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const f32x2_t xh = y[k] * rs2;
            const f32x2_t z = __builtin_elementwise_fma(xh, g2[k], b2[k]);
            f32x2_t a, gz;
            if constexpr (sizeof(T) == 2) {        // bf16 build: fast erf, one exp shared by gelu and gelu' (common.h)
                gelu_both2(z[0], z[1], a, gz);
            } else {
                a = (f32x2_t){gelu_f(z[0]), gelu_f(z[1])};
                gz = (f32x2_t){gelu_grad_f(z[0]), gelu_grad_f(z[1])};
            }
            const f32x2_t da = __builtin_elementwise_fma(dd2, w2[2][k], __builtin_elementwise_fma(dd1, w2[1][k], dd0 * w2[0][k]));
            const f32x2_t dz = da * gz;
            ag[k] = __builtin_elementwise_fma(dz, xh, ag[k]);
            ab[k] = ab[k] + dz;
            aw[0][k] = __builtin_elementwise_fma(dd0, a, aw[0][k]);
            aw[1][k] = __builtin_elementwise_fma(dd1, a, aw[1][k]);
            aw[2][k] = __builtin_elementwise_fma(dd2, a, aw[2][k]);
            dxh[k] = dz * g2[k];
            m1 = m1 + dxh[k];
            m2 = __builtin_elementwise_fma(dxh[k], xh, m2);
            y[k] = xh;
        }
        if (sub == 0) { ab1[0] += d0; ab1[1] += d1; ab1[2] += d2; }
        const float m1s = oct_sum(m1[0] + m1[1]) * (1.f / 64), m2s = oct_sum(m2[0] + m2[1]) * (1.f / 64);
        const f32x2_t mm1 = {m1s, m1s}, nm2 = {-m2s, -m2s};
        f32x2_t r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = rs2 * (__builtin_elementwise_fma(y[k], nm2, dxh[k]) - mm1);
        if constexpr (sizeof(T) == 2) {
            *reinterpret_cast<uint4*>(dy3 + (size_t)pix * 64 + c0) = make_uint4(pack_bf16x2(r[0][0], r[0][1]), pack_bf16x2(r[1][0], r[1][1]),
                                                                                 pack_bf16x2(r[2][0], r[2][1]), pack_bf16x2(r[3][0], r[3][1]));
        } else {
            *reinterpret_cast<float4*>(dy3 + (size_t)pix * 64 + c0) = make_float4(r[0][0], r[0][1], r[1][0], r[1][1]);
            *reinterpret_cast<float4*>(dy3 + (size_t)pix * 64 + c0 + 4) = make_float4(r[2][0], r[2][1], r[3][0], r[3][1]);
        }
    }
    // the 8 lanes l, l + 8, ..., l + 56 hold the same 8 channels (other pixels): rotate-by-8 inside a row of 16 lanes, then across rows
    auto wsum = [&](float v) { v += lane_dpp<0x128>(v); v += lane_xor16(v); v += lane_xor32(v); return v; };
    __shared__ float red[4][TAILP];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float vg = wsum(ag[k][e]), vb = wsum(ab[k][e]), v0 = wsum(aw[0][k][e]), v1 = wsum(aw[1][k][e]), v2 = wsum(aw[2][k][e]);
            if (lane < 8) {
                const int c = c0 + 2 * k + e;
                red[wave][c] = vg; red[wave][64 + c] = vb; red[wave][128 + c] = v0; red[wave][192 + c] = v1; red[wave][256 + c] = v2;
            }
        }
    {
        const float t0 = wsum(ab1[0]), t1 = wsum(ab1[1]), t2 = wsum(ab1[2]);      // (only the sub == 0 lanes carry values: lanes 0, 8, ..., 56)
        if (lane == 0) { red[wave][320] = t0; red[wave][321] = t1; red[wave][322] = t2; red[wave][323] = 0.f; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < TAILP; i += 256) part[(size_t)blockIdx.x * TAILP + i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
}
static int tail_bwd_blocks(int npix) {
    int b = (npix + 31) / 32;
    return b > 2048 ? 2048 : b;
}
extern "C" int64_t pa_decoder_tail_bwd_workspace_bytes(int batch, int Hi, int Wi) { return (int64_t)tail_bwd_blocks(batch * Hi * Wi) * TAILP * sizeof(float); }
// grads: f32 [324] = dgamma[64] | dbeta[64] | dW1[3*64] | db1[3] | pad
extern "C" int pa_decoder_tail_bwd_pointwise(int dtype, const float* dpred, const void* y3, const float* ln_gamma, const float* ln_beta,
                                             const float* w1, void* dy3, float* grads, void* workspace, int batch, int Hi, int Wi, float eps,
                                             hipStream_t st) {
    const int npix = batch * Hi * Wi, nb = tail_bwd_blocks(npix);
    float* part = reinterpret_cast<float*>(workspace);
    if (dtype == PA_BF16)
        PA_LAUNCH(tail_bwd_kernel<bf16>, dim3(nb), dim3(256), 0, st, dpred, (const bf16*)y3, ln_gamma, ln_beta, w1, (bf16*)dy3, part, Hi * Wi, npix, eps);
    else
        PA_LAUNCH(tail_bwd_kernel<float>, dim3(nb), dim3(256), 0, st, dpred, (const float*)y3, ln_gamma, ln_beta, w1, (float*)dy3, part, Hi * Wi, npix, eps);
    int e = (int)hipGetLastError();
    if (e) return e;
    return pa_slab_reduce(part, grads, TAILP, nb, TAILP, 0, st);
}

// ------------------------------------------------------------------------------- diagnostics: a stand-in for a collective's kernel
// scratch[i] = 0.5 * (scratch[i] + src[i]), `passes` times, on exactly `nblocks` persistent workgroups of 256 threads: the HBM traffic
// (12 B per element per pass) and CU footprint of a ring all-reduce step with `nblocks` channels, without touching the gradient it
// reads.  Used by tools/gradsync_overlap.py to price the backward's loss of CUs / bandwidth to RCCL on a single-GPU box, where a
// 1-rank group launches no ring kernel at all.  Never on the product path.
__global__ __launch_bounds__(256) void debug_rmw_kernel(const float* __restrict__ src, float* __restrict__ scratch, size_t n4, int passes) {
    for (int p = 0; p < passes; ++p)
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
            const float4 a = reinterpret_cast<const float4*>(src)[i];
            float4 b = reinterpret_cast<float4*>(scratch)[i];
            b.x = 0.5f * (a.x + b.x); b.y = 0.5f * (a.y + b.y); b.z = 0.5f * (a.z + b.z); b.w = 0.5f * (a.w + b.w);
            reinterpret_cast<float4*>(scratch)[i] = b;
        }
}
extern "C" int pa_debug_rmw(const float* src, float* scratch, int64_t n, int passes, int nblocks, hipStream_t st) {
    if (n % 4 || nblocks < 1 || passes < 1) return (int)hipErrorInvalidValue;
    PA_LAUNCH(debug_rmw_kernel, dim3(nblocks), dim3(256), 0, st, src, scratch, (size_t)(n / 4), passes);
    LAUNCH_CHECK();
}
