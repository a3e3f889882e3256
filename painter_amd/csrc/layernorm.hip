// LayerNorm forward / backward over the channel dim of the fp32 residual stream.
// Replaces aten::native_layer_norm(_backward) at Painter/models_painter.py:218,230 (norm1/norm2,
// eps 1e-6 via partial(nn.LayerNorm) :480) and the shared tap norm :416-417.
// HBM-bound: one wave per row, 16-byte loads, two-pass variance in registers, shuffle reductions.
#include "common.h"
#include "../../include/painter_hip.h"

template <typename T> DEVI void store4(T* p, float a, float b, float c, float d);
template <> DEVI void store4<float>(float* p, float a, float b, float c, float d) {
    *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
}
template <> DEVI void store4<bf16>(bf16* p, float a, float b, float c, float d) {
    *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16x2(a, b), pack_bf16x2(c, d));
}
DEVI float4 load4(const float* p) { return *reinterpret_cast<const float4*>(p); }
DEVI float4 load4(const bf16* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    return make_float4(bf16_lo(u.x), bf16_hi(u.x), bf16_lo(u.y), bf16_hi(u.y));
}
template <typename T, int NI>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, size_t ldx, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float eps, T* __restrict__ y, size_t ldy,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int R, int D) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= R) return;
    const float* xr = x + (size_t)row * ldx;
    float4 v[NI];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int c = lane * 4 + 256 * i;
        v[i] = (c < D) ? load4(xr + c) : make_float4(0, 0, 0, 0);
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mu = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int c = lane * 4 + 256 * i;
        if (c < D) {
            const float a = v[i].x - mu, b = v[i].y - mu, cc = v[i].z - mu, d = v[i].w - mu;
            q += (a * a + b * b) + (cc * cc + d * d);
        }
    }
    const float rs = rsqrtf(wave_sum(q) / (float)D + eps);
    if (lane == 0) {
        mean[row] = mu;
        rstd[row] = rs;
    }
    T* yr = y + (size_t)row * ldy;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int c = lane * 4 + 256 * i;
        if (c < D) {
            const float4 g = load4(gamma + c), b = load4(beta + c);
            store4<T>(yr + c, (v[i].x - mu) * rs * g.x + b.x, (v[i].y - mu) * rs * g.y + b.y,
                      (v[i].z - mu) * rs * g.z + b.z, (v[i].w - mu) * rs * g.w + b.w);
        }
    }
}

// dx_out = (dres ? dres : 0) + LNbwd(dy);  optional T copy dxT = rowscale[row / rps] * dx_out;
// per-workgroup partial dgamma / dbeta -> part[block][NP][D], NP = 2, or 3 with CS: the third row is the column sum of rowscale * dx (the fp32
// values dxT is rounded from) = the bias gradient of the nn.Linear whose dY this dxT is (fc2 / proj) -- it used to be a separate pass over dxT
// (4 waves per SIMD: with the column-sum accumulators the kernel asked for 130 VGPRs = 3 waves per SIMD and ran 40 % slower, 69.6 vs 49.8 us
// per ViT-L launch -- this HBM-bound stream needs the fourth wave to keep enough loads in flight)
template <typename T, int NI, bool CS>
__global__ __launch_bounds__(256, NI <= 4 ? 4 : 2) void ln_bwd_kernel(const T* __restrict__ dy, size_t lddy, const float* __restrict__ x, size_t ldx,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, const float* dres, float* dx, size_t lddx,
                                                     T* dxT, size_t lddxT, const float* __restrict__ rowscale, int rps,
                                                     float* __restrict__ part, int R, int D) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NP = CS ? 3 : 2;
    float4 g[NI], ag[NI], ab[NI], ac[CS ? NI : 1];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int c = lane * 4 + 256 * i;
        g[i] = (c < D) ? load4(gamma + c) : make_float4(0, 0, 0, 0);
        ag[i] = make_float4(0, 0, 0, 0);
        ab[i] = make_float4(0, 0, 0, 0);
        if constexpr (CS) ac[i] = make_float4(0, 0, 0, 0);
    }
    for (int row = blockIdx.x * 4 + wave; row < R; row += gridDim.x * 4) {
        const float mu = mean[row], rs = rstd[row];
        const T* dyr = dy + (size_t)row * lddy;
        const float* xr = x + (size_t)row * ldx;
        float4 d[NI], xh[NI];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int c = lane * 4 + 256 * i;
            if (c < D) {
                d[i] = load4(dyr + c);
                const float4 xv = load4(xr + c);
                xh[i] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
                ag[i].x += d[i].x * xh[i].x; ag[i].y += d[i].y * xh[i].y; ag[i].z += d[i].z * xh[i].z; ag[i].w += d[i].w * xh[i].w;
                ab[i].x += d[i].x; ab[i].y += d[i].y; ab[i].z += d[i].z; ab[i].w += d[i].w;
                d[i].x *= g[i].x; d[i].y *= g[i].y; d[i].z *= g[i].z; d[i].w *= g[i].w;      // dxhat
                s1 += (d[i].x + d[i].y) + (d[i].z + d[i].w);
                s2 += (d[i].x * xh[i].x + d[i].y * xh[i].y) + (d[i].z * xh[i].z + d[i].w * xh[i].w);
            }
        }
        const float c1 = wave_sum(s1) / (float)D, c2 = wave_sum(s2) / (float)D;
        const float sc = (dxT && rowscale) ? rowscale[row / rps] : 1.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int c = lane * 4 + 256 * i;
            if (c < D) {
                float4 o = make_float4(rs * (d[i].x - c1 - xh[i].x * c2), rs * (d[i].y - c1 - xh[i].y * c2),
                                       rs * (d[i].z - c1 - xh[i].z * c2), rs * (d[i].w - c1 - xh[i].w * c2));
                if (dres) {
                    const float4 r = load4(dres + (size_t)row * lddx + c);
                    o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
                }
                *reinterpret_cast<float4*>(dx + (size_t)row * lddx + c) = o;
                if (dxT) store4<T>(dxT + (size_t)row * lddxT + c, o.x * sc, o.y * sc, o.z * sc, o.w * sc);
                if constexpr (CS) {          // fp32 values in front of dxT's rounding (a round trip through T per element made the kernel VALU-heavier for nothing)
                    ac[i].x = fmaf(o.x, sc, ac[i].x); ac[i].y = fmaf(o.y, sc, ac[i].y);
                    ac[i].z = fmaf(o.z, sc, ac[i].z); ac[i].w = fmaf(o.w, sc, ac[i].w);
                }
            }
        }
    }
    // block reduce of the 4 waves' partial sums, through [4][2][D] floats of LDS whatever NP is (the column sum goes through the same
    // 32 KB in a second round: with [4][3][D] the kernel dropped from 4 to 3 resident workgroups per CU)
    extern __shared__ float red[];   // [4][2][D]
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int c = lane * 4 + 256 * i;
        if (c < D) {
            *reinterpret_cast<float4*>(red + (size_t)(wave * 2 + 0) * D + c) = ag[i];
            *reinterpret_cast<float4*>(red + (size_t)(wave * 2 + 1) * D + c) = ab[i];
        }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 2 * D; idx += 256) {
        const float s = (red[idx] + red[2 * D + idx]) + (red[4 * D + idx] + red[6 * D + idx]);
        part[(size_t)blockIdx.x * NP * D + idx] = s;
    }
    if constexpr (CS) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int c = lane * 4 + 256 * i;
            if (c < D) *reinterpret_cast<float4*>(red + (size_t)wave * D + c) = ac[i];
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < D; idx += 256) {
            const float s = (red[idx] + red[D + idx]) + (red[2 * D + idx] + red[3 * D + idx]);
            part[(size_t)blockIdx.x * NP * D + 2 * D + idx] = s;
        }
    }
}

// ---- round 5: the same backward with each ROW split over the workgroup's 4 waves (D >= 1024) ----------------------------------------------
// One wave per row kept 64 registers of parameter-gradient accumulators per lane (gamma, d-gamma, d-beta, column sum: 4 float4 each at
// D = 1024) next to the row itself, ran its two memory round trips (dy / x, then the residual gradient) back to back and had no registers
// left to start the next row: 4.1 TB/s where the stream could go faster (waves waiting on memory 86 % of the time, PMC).  Here thread t
// owns columns 4t .. 4t+3 (+ 1024 j) of EVERY row its workgroup handles: 4 accumulators of one float4, RB (= 2) rows per batch in registers, the
// next batch's loads -- residual gradient included -- issued before the current one is reduced, row statistics combined through 128 bytes
// of LDS (one barrier per batch, two slots), and no cross-wave reduction of the partial rows at the end: every column has one owner.
// Batches go to workgroups in contiguous, equal runs (G = ceil(batches / k) <= the resident capacity), so nobody waits for a second round.
DEVI uint2 raw4(const bf16* p) { return *reinterpret_cast<const uint2*>(p); }
DEVI float4 raw4(const float* p) { return *reinterpret_cast<const float4*>(p); }
DEVI float4 cvt4(uint2 u) { return make_float4(bf16_lo(u.x), bf16_hi(u.x), bf16_lo(u.y), bf16_hi(u.y)); }
DEVI float4 cvt4(float4 v) { return v; }
template <typename T> struct Raw4 { typedef float4 type; };
template <> struct Raw4<bf16> { typedef uint2 type; };

template <typename T, int NJ, int RB, int OCC, bool CS>
__global__ __launch_bounds__(256, OCC) void ln_bwd_rows_kernel(const T* __restrict__ dy, size_t lddy, const float* __restrict__ x, size_t ldx,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          const float* __restrict__ gamma, const float* dres, float* dx, size_t lddx,
                                                          T* dxT, size_t lddxT, const float* __restrict__ rowscale, int rps,
                                                          float* __restrict__ part, int R, int D, int nbatch) {
    typedef typename Raw4<T>::type RawT;
    constexpr int NP = CS ? 3 : 2;
    __shared__ float red[2][4][RB][2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b0 = (int)((long long)blockIdx.x * nbatch / gridDim.x), b1 = (int)((long long)(blockIdx.x + 1) * nbatch / gridDim.x);
    int col[NJ];
    bool on[NJ];
    float4 g[NJ], ag[NJ], ab[NJ], ac[CS ? NJ : 1];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        col[j] = tid * 4 + 1024 * j;
        on[j] = col[j] < D;
        g[j] = on[j] ? load4(gamma + col[j]) : make_float4(0, 0, 0, 0);
        ag[j] = make_float4(0, 0, 0, 0);
        ab[j] = make_float4(0, 0, 0, 0);
        if constexpr (CS) ac[j] = make_float4(0, 0, 0, 0);
    }
    const float invD = 1.f / (float)D;
    RawT ndy[RB][NJ];
    float4 nx[RB][NJ], nr[RB][NJ];
    float nmu[RB], nrs[RB], nsc[RB];
    auto fetch = [&](int b) {
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int row = min(b * RB + r, R - 1);              // a clamped (repeated) last row is loaded and dropped
            nmu[r] = mean[row];
            nrs[r] = rstd[row];
            nsc[r] = (dxT && rowscale) ? rowscale[row / rps] : 1.f;
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                if (on[j]) {
                    ndy[r][j] = raw4(dy + (size_t)row * lddy + col[j]);
                    nx[r][j] = load4(x + (size_t)row * ldx + col[j]);
                    if (dres) nr[r][j] = load4(dres + (size_t)row * lddx + col[j]);
                }
        }
    };
    if (b0 < b1) fetch(b0);
    int par = 0;
    for (int b = b0; b < b1; ++b) {
        RawT cdy[RB][NJ];
        float4 cx[RB][NJ], cr[RB][NJ];
        float mu[RB], rs[RB], sc[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            mu[r] = nmu[r]; rs[r] = nrs[r]; sc[r] = nsc[r];
#pragma unroll
            for (int j = 0; j < NJ; ++j) { cdy[r][j] = ndy[r][j]; cx[r][j] = nx[r][j]; cr[r][j] = nr[r][j]; }
        }
        if (b + 1 < b1) fetch(b + 1);
        float4 d[RB][NJ], xh[RB][NJ];
        float s1[RB], s2[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const bool live = b * RB + r < R;                    // workgroup-uniform
            s1[r] = 0.f;
            s2[r] = 0.f;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                d[r][j] = make_float4(0, 0, 0, 0);
                xh[r][j] = make_float4(0, 0, 0, 0);
                if (on[j] && live) {
                    const float4 dv = cvt4(cdy[r][j]), xv = cx[r][j];
                    const float4 h = make_float4((xv.x - mu[r]) * rs[r], (xv.y - mu[r]) * rs[r], (xv.z - mu[r]) * rs[r], (xv.w - mu[r]) * rs[r]);
                    ag[j].x += dv.x * h.x; ag[j].y += dv.y * h.y; ag[j].z += dv.z * h.z; ag[j].w += dv.w * h.w;
                    ab[j].x += dv.x; ab[j].y += dv.y; ab[j].z += dv.z; ab[j].w += dv.w;
                    const float4 q = make_float4(dv.x * g[j].x, dv.y * g[j].y, dv.z * g[j].z, dv.w * g[j].w);      // dxhat
                    s1[r] += (q.x + q.y) + (q.z + q.w);
                    s2[r] += (q.x * h.x + q.y * h.y) + (q.z * h.z + q.w * h.w);
                    d[r][j] = q;
                    xh[r][j] = h;
                }
            }
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            s1[r] = wave_sum(s1[r]);
            s2[r] = wave_sum(s2[r]);
        }
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                red[par][wave][r][0] = s1[r];
                red[par][wave][r][1] = s2[r];
            }
        }
        __syncthreads();           // a thread is at most one barrier ahead of another: batch b + 1 writes the other slot, batch b + 2 comes after every read of b
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int row = b * RB + r;
            if (row >= R) break;
            const float c1 = ((red[par][0][r][0] + red[par][1][r][0]) + (red[par][2][r][0] + red[par][3][r][0])) * invD;
            const float c2 = ((red[par][0][r][1] + red[par][1][r][1]) + (red[par][2][r][1] + red[par][3][r][1])) * invD;
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                if (on[j]) {
                    const float4 q = d[r][j], h = xh[r][j];
                    float4 o = make_float4(rs[r] * (q.x - c1 - h.x * c2), rs[r] * (q.y - c1 - h.y * c2),
                                           rs[r] * (q.z - c1 - h.z * c2), rs[r] * (q.w - c1 - h.w * c2));
                    if (dres) { o.x += cr[r][j].x; o.y += cr[r][j].y; o.z += cr[r][j].z; o.w += cr[r][j].w; }
                    *reinterpret_cast<float4*>(dx + (size_t)row * lddx + col[j]) = o;
                    if (dxT) store4<T>(dxT + (size_t)row * lddxT + col[j], o.x * sc[r], o.y * sc[r], o.z * sc[r], o.w * sc[r]);
                    if constexpr (CS) {
                        ac[j].x = fmaf(o.x, sc[r], ac[j].x); ac[j].y = fmaf(o.y, sc[r], ac[j].y);
                        ac[j].z = fmaf(o.z, sc[r], ac[j].z); ac[j].w = fmaf(o.w, sc[r], ac[j].w);
                    }
                }
        }
        par ^= 1;
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j)
        if (on[j]) {
            float* pr = part + (size_t)blockIdx.x * NP * D + col[j];
            *reinterpret_cast<float4*>(pr) = ag[j];
            *reinterpret_cast<float4*>(pr + D) = ab[j];
            if constexpr (CS) *reinterpret_cast<float4*>(pr + 2 * D) = ac[j];
        }
}

extern "C" int pa_slab_reduce(const float* in, float* out, int64_t n, int nz, int64_t stride, int accumulate, hipStream_t st);
int pa_slab_reduce2(const float* in, float* out0, float* out1, int64_t n0, int64_t n, int nz, int64_t stride, hipStream_t st);   // gemm.hip

template <typename T>
static int ln_fwd_t(const float* x, int64_t ldx, const float* g, const float* b, float eps, T* y, int64_t ldy, float* mean,
                    float* rstd, int R, int D, hipStream_t st) {
    const int ni = (D + 255) / 256;
    dim3 grid((R + 3) / 4), blk(256);
#define LN_FWD(NI) PA_LAUNCH((ln_fwd_kernel<T, NI>), grid, blk, 0, st, x, (size_t)ldx, g, b, eps, y, (size_t)ldy, mean, rstd, R, D)
    if (ni <= 1) LN_FWD(1);
    else if (ni <= 2) LN_FWD(2);
    else if (ni <= 4) LN_FWD(4);
    else if (ni <= 5) LN_FWD(5);
    else if (ni <= 8) LN_FWD(8);
    else return (int)hipErrorInvalidValue;
#undef LN_FWD
    LAUNCH_CHECK();
}
extern "C" int pa_layernorm_fwd(int dtype, const float* x, int64_t ldx, const float* gamma, const float* beta, float eps,
                                void* y, int64_t ldy, float* mean, float* rstd, int R, int D, hipStream_t st) {
    if (D % 4 || ldx % 4 || ldy % 4) return (int)hipErrorInvalidValue;
    if (dtype == PA_BF16) return ln_fwd_t<bf16>(x, ldx, gamma, beta, eps, (bf16*)y, ldy, mean, rstd, R, D, st);
    return ln_fwd_t<float>(x, ldx, gamma, beta, eps, (float*)y, ldy, mean, rstd, R, D, st);
}

static int ln_bwd_blocks_wave(int R) {
    int b = (R + 3) / 4;
    return b > 1024 ? 1024 : b;      // 4 waves per SIMD resident on 256 CUs
}
// row-split kernel: RB rows per batch, OCC workgroups resident per CU
// batches of 2 rows; 4 workgroups per CU at D <= 1024 (98 registers), 3 with two column groups per thread (1024 < D <= 2048).  Measured in
// the ViT-L step against one wave per row (profiles/r05_ab_layernorm_bwd_rows_split.log): 4-row batches / 3 per CU -0.59 ms, 2-row batches /
// 4 per CU -0.68 ms, 2-row batches / 5 per CU +0.15 ms (6 spilled registers and 1255 short workgroups).
constexpr int LNB_RB = 2;
static int lnb_occ(int D) { return D <= 1024 ? 4 : 3; }
static bool ln_bwd_rows_ok(int D) { return g_ln_bwd_variant != 1 && D >= 1024 && D <= 2048; }
static int ln_bwd_blocks_rows(int R, int D) {
    const int nbatch = (R + LNB_RB - 1) / LNB_RB, cap = 256 * lnb_occ(D);
    if (nbatch <= 0) return 0;                                  // an empty batch: no workgroups, no workspace (and no division by k = 0)
    const int k = (nbatch + cap - 1) / cap;                     // batches per workgroup
    return (nbatch + k - 1) / k;
}
static int ln_bwd_blocks(int R, int D) { return ln_bwd_rows_ok(D) ? ln_bwd_blocks_rows(R, D) : ln_bwd_blocks_wave(R); }
// (sized for either variant: the knob may change between the call that sized a caller-owned buffer and the launch)
extern "C" int64_t pa_layernorm_bwd_workspace_bytes(int R, int D) {
    const int a = ln_bwd_blocks_wave(R), b = ln_bwd_blocks_rows(R, D);
    return (int64_t)(a > b ? a : b) * 3 * D * sizeof(float);
}

template <typename T>
static int ln_bwd_t(const T* dy, int64_t lddy, const float* x, int64_t ldx, const float* mean, const float* rstd,
                    const float* gamma, const float* dres, float* dx, int64_t lddx, T* dxT, int64_t lddxT,
                    const float* rowscale, int rps, float* dgamma_dbeta, float* dxT_colsum, float* ws, int R, int D, hipStream_t st) {
    const int ni = (D + 255) / 256;
    const int nb = ln_bwd_blocks(R, D);
    dim3 grid(nb), blk(256);
    const bool cs = dxT_colsum != nullptr;       // (with dgamma_dbeta == NULL: any non-NULL value selects the column-sum partials)
    if (cs && dxT == nullptr) return (int)hipErrorInvalidValue;
    const int np = cs ? 3 : 2;
    if (ln_bwd_rows_ok(D)) {
        const int nbatch = (R + LNB_RB - 1) / LNB_RB;
#define LN_ROWS(NJ, OCC_, CS_) PA_LAUNCH((ln_bwd_rows_kernel<T, NJ, LNB_RB, OCC_, CS_>), grid, blk, 0, st, dy, (size_t)lddy, x, (size_t)ldx, mean, rstd, gamma, dres, dx, (size_t)lddx, dxT, (size_t)lddxT, rowscale, rps, ws, R, D, nbatch)
        if (D <= 1024) { if (cs) LN_ROWS(1, 4, true); else LN_ROWS(1, 4, false); }
        else { if (cs) LN_ROWS(2, 3, true); else LN_ROWS(2, 3, false); }
#undef LN_ROWS
        int e = (int)hipGetLastError();
        if (e || dgamma_dbeta == nullptr) return e;
        return pa_slab_reduce2(ws, dgamma_dbeta, dxT_colsum, 2 * D, np * D, nb, np * D, st);
    }
    const size_t sm = (size_t)8 * D * sizeof(float);          // <= 64 KB for every D the kernel takes (NI <= 8: D <= 2048)
    // (round 5: non-temporal loads of the read-once streams -- x saved by the forward, dy from the GEMM in front -- measured 53.11 vs 53.25 ms
    // per step, inside the noise: profiles/r05_ab_layernorm_bwd_nt_loads.log; not kept)
#define LN_BWD2(NI, CS_) PA_LAUNCH((ln_bwd_kernel<T, NI, CS_>), grid, blk, sm, st, dy, (size_t)lddy, x, (size_t)ldx, mean, rstd, gamma, dres, dx, (size_t)lddx, dxT, (size_t)lddxT, rowscale, rps, ws, R, D)
#define LN_BWD(NI) do { if (cs) LN_BWD2(NI, true); else LN_BWD2(NI, false); } while (0)
    if (ni <= 1) LN_BWD(1);
    else if (ni <= 2) LN_BWD(2);
    else if (ni <= 4) LN_BWD(4);
    else if (ni <= 5) LN_BWD(5);
    else if (ni <= 8) LN_BWD(8);
    else return (int)hipErrorInvalidValue;
#undef LN_BWD
#undef LN_BWD2
    int e = (int)hipGetLastError();
    if (e || dgamma_dbeta == nullptr) return e;        // NULL: the caller reduces the partial rows later (pa_layernorm_bwd_reduce, e.g. on another stream)
    // partial rows are [dgamma | dbeta (| colsum)]: one reduction launch writes the first 2 D sums to dgamma_dbeta and the rest to dxT_colsum
    return pa_slab_reduce2(ws, dgamma_dbeta, dxT_colsum, 2 * D, np * D, nb, np * D, st);
}
// the reduction of pa_layernorm_bwd(dgamma_dbeta = NULL, ...)'s partial rows; with_colsum says whether that call was given a dxT_colsum
extern "C" int pa_layernorm_bwd_reduce(const void* workspace, float* dgamma_dbeta, float* dxT_colsum, int with_colsum, int R, int D, hipStream_t st) {
    if (R <= 0 || workspace == nullptr || dgamma_dbeta == nullptr || (with_colsum && dxT_colsum == nullptr)) return (int)hipErrorInvalidValue;
    const int np = with_colsum ? 3 : 2;
    return pa_slab_reduce2((const float*)workspace, dgamma_dbeta, dxT_colsum, 2 * D, np * D, ln_bwd_blocks(R, D), np * D, st);
}
// dgamma_dbeta: [2, D] fp32 (dgamma then dbeta), overwritten.
extern "C" int pa_layernorm_bwd(int dtype, const void* dy, int64_t lddy, const float* x, int64_t ldx, const float* mean,
                                const float* rstd, const float* gamma, const float* dres, float* dx, int64_t lddx,
                                void* dxT, int64_t lddxT, const float* rowscale, int rows_per_sample,
                                float* dgamma_dbeta, float* dxT_colsum, void* workspace, int R, int D, hipStream_t st) {
    if (R <= 0 || D % 4 || lddy % 4 || ldx % 4 || lddx % 4 || lddxT % 4) return (int)hipErrorInvalidValue;
    if (dtype == PA_BF16)
        return ln_bwd_t<bf16>((const bf16*)dy, lddy, x, ldx, mean, rstd, gamma, dres, dx, lddx, (bf16*)dxT, lddxT, rowscale,
                              rows_per_sample, dgamma_dbeta, dxT_colsum, (float*)workspace, R, D, st);
    return ln_bwd_t<float>((const float*)dy, lddy, x, ldx, mean, rstd, gamma, dres, dx, lddx, (float*)dxT, lddxT, rowscale,
                           rows_per_sample, dgamma_dbeta, dxT_colsum, (float*)workspace, R, D, st);
}
