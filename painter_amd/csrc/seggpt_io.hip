// SegGPT pre-/post-processing on the device: everything seggpt_engine.py does around the model call with PIL,
// numpy and CPU torch (SegGPT/SegGPT_inference/seggpt_engine.py:56-103, :106-181, :26-53), as byte / index kernels.
//
// All of it is HBM- and launch-bound integer / byte work (a 1080p frame is 6 MB, the 448 x 448 pictures 0.6 MB), so the kernels are
// plain coalesced one-thread-per-output-pixel gathers: rows of the output map to consecutive lanes, every byte of the big frame is
// read once and written once, the small operands (coefficient / index tables, the 1568 x 768 token matrix) stay in L2.  No MFMA.
// LDS only where there is reuse to capture: the horizontal resize pass, whose ~19 taps per output pixel overlap 4x along the row,
// stages each source row once.
//
// Bit-exactness with the reference's host path is the contract:
//   * resize passes: Pillow's 8-bit fixed point (int32 accumulator seeded with 2^21, >> 22, clamp) -- pure integer;
//   * normalise / de-normalise / blend: float64 with ONE rounding per operation, in the reference's operation order.  This file is
//     compiled with -ffp-contract=off (build.py) and carries the pragma below: an fma in `(y * std + mean)` or `0.6 * o / 255 + 0.4`
//     changes the last bit, and the final uint8 truncation turns that into an off-by-one for saturated pixels
//     (200 * 0.9999999999999999 -> 199).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/painter_hip.h"
#include "common.h"

#pragma clang fp contract(off)

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;          // Pillow Resample.c
constexpr int MAXC = 4;

__device__ __constant__ double kMean[3] = {0.485, 0.456, 0.406};      // seggpt_engine.py:9
__device__ __constant__ double kStd[3] = {0.229, 0.224, 0.225};       // seggpt_engine.py:10

DEVI uint8_t clip8(int acc) {
    const int v = acc >> PRECISION_BITS;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// grid: x = blocks of 256 output columns, y = output rows.  Horizontal: taps walk along the row (3 B apart, served by L1/L2);
// vertical: taps walk down the rows, lanes stay on consecutive columns.
template <bool VERT>
__global__ __launch_bounds__(256) void resample_u8_kernel(const uint8_t* __restrict__ src, size_t row_bytes, uint8_t* __restrict__ dst, int dw, int C,
                                                          const int* __restrict__ bounds, const int* __restrict__ coeffs, int ksize) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= dw) return;
    const int o = VERT ? y : x;
    const int first = bounds[2 * o], taps = bounds[2 * o + 1];
    const int* k = coeffs + (size_t)o * ksize;
    int acc[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) acc[c] = 1 << (PRECISION_BITS - 1);
    const uint8_t* s = VERT ? src + (size_t)first * row_bytes + (size_t)x * C : src + (size_t)y * row_bytes + (size_t)first * C;
    const size_t step = VERT ? row_bytes : (size_t)C;
    for (int t = 0; t < taps; ++t, s += step) {
        const int kt = k[t];
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (c < C) acc[c] += (int)s[c] * kt;
    }
    uint8_t* d = dst + ((size_t)y * dw + x) * C;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
        if (c < C) d[c] = clip8(acc[c]);
}

// Horizontal pass, one workgroup per input row: the row (a few KB) is staged in LDS with aligned dword loads -- every source byte
// crosses HBM/L2 once, coalesced -- and the taps are LDS byte reads.  The direct kernel above makes ~57 byte-granular global loads
// per output pixel (30.6 us for 1080p -> 448 columns); this one is bounded by the 6 MB read.  LDS image keeps the row's misalignment
// (off = address & 3) so that global dword i lands on an aligned LDS dword.
__global__ __launch_bounds__(256) void resample_h_lds_kernel(const uint8_t* __restrict__ src, size_t row_bytes, int sw, uint8_t* __restrict__ dst, int dw, int C,
                                                             const int* __restrict__ bounds, const int* __restrict__ coeffs, int ksize) {
    extern __shared__ __attribute__((aligned(16))) unsigned char row_lds[];
    const int y = blockIdx.x, tid = threadIdx.x;
    const int nbytes = sw * C;
    const uint8_t* row = src + (size_t)y * row_bytes;
    const int off = (int)((uintptr_t)row & 3);
    int head = (4 - off) & 3;
    if (head > nbytes) head = nbytes;
    const int ndw = (nbytes - head) >> 2, tail0 = head + 4 * ndw;
    if (tid < head) row_lds[off + tid] = row[tid];
    const uint32_t* rp = (const uint32_t*)(row + head);
    uint32_t* lp = (uint32_t*)(row_lds + off + head);
    for (int i = tid; i < ndw; i += 256) lp[i] = rp[i];
    if (tid < nbytes - tail0) row_lds[off + tail0 + tid] = row[tail0 + tid];
    __syncthreads();
    const unsigned char* r = row_lds + off;
    for (int x = tid; x < dw; x += 256) {
        const int first = bounds[2 * x], taps = bounds[2 * x + 1];
        const int* k = coeffs + (size_t)x * ksize;
        int acc[MAXC];
#pragma unroll
        for (int c = 0; c < MAXC; ++c) acc[c] = 1 << (PRECISION_BITS - 1);
        const unsigned char* s = r + first * C;
        for (int t = 0; t < taps; ++t, s += C) {
            const int kt = k[t];
#pragma unroll
            for (int c = 0; c < MAXC; ++c)
                if (c < C) acc[c] += (int)s[c] * kt;
        }
        uint8_t* d = dst + ((size_t)y * dw + x) * C;
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (c < C) d[c] = clip8(acc[c]);
    }
}

__global__ __launch_bounds__(256) void gather_u8_kernel(const uint8_t* __restrict__ src, size_t row_bytes, uint8_t* __restrict__ dst, int dw, int C,
                                                        const int* __restrict__ ytab, const int* __restrict__ xtab) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= dw) return;
    const int sy = ytab[y], sx = xtab[x];
    uint8_t* d = dst + ((size_t)y * dw + x) * C;
    if (sy < 0 || sx < 0) {
        for (int c = 0; c < C; ++c) d[c] = 0;
        return;
    }
    const uint8_t* s = src + (size_t)sy * row_bytes + (size_t)sx * C;
    for (int c = 0; c < C; ++c) d[c] = s[c];
}

// (v / div - mean) / std, one rounding per operation, then narrowed to float32 (numpy float64 -> torch .float()).
DEVI float normalise(uint8_t u, double div, int c) {
    double v = (double)u / div;
    v = v - kMean[c];
    v = v / kStd[c];
    return (float)v;
}

// grid: x = blocks of 256 columns, y = row of the stitched 2R x W canvas, z = prompt.  Reads 3-byte pixels, writes three coalesced
// float32 planes for imgs and tgts.
__global__ __launch_bounds__(256) void stitch_kernel(const uint8_t* __restrict__ prompts, const uint8_t* __restrict__ targets,
                                                     const double* __restrict__ target_div, const uint8_t* __restrict__ query,
                                                     float* __restrict__ imgs, float* __restrict__ tgts, int R, int W) {
    const int x = blockIdx.x * 256 + threadIdx.x, row = blockIdx.y, n = blockIdx.z;
    if (x >= W) return;
    const int r = row < R ? row : row - R;
    const size_t px = ((size_t)r * W + x) * 3, img_sz = (size_t)R * W * 3;
    const uint8_t* a = row < R ? prompts + n * img_sz + px : query + px;
    const uint8_t* t = targets + n * img_sz + px;
    const double div = target_div[n];
    const size_t plane = (size_t)2 * R * W;
    const size_t o = (size_t)n * 3 * plane + (size_t)row * W + x;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        imgs[o + c * plane] = normalise(a[c], 255.0, c);
        tgts[o + c * plane] = normalise(t[c], div, c);
    }
}

// Element (r, x, c) of the LOWER res_h x res_w half of unpatchify(pred) (models_seggpt.py:376-389), then seggpt_engine.py:52:
// clip((v * std + mean) * 255, 0, 255) in float64.
DEVI double decoded(const float* __restrict__ pred, int r, int x, int c, int res_h, int wp, int P) {
    const int row = res_h + r;
    const int token = (row / P) * wp + x / P;
    const int within = ((row % P) * P + x % P) * 3 + c;
    double o = (double)pred[(size_t)token * (P * P * 3) + within];
    o = o * kStd[c];
    o = o + kMean[c];
    o = o * 255.0;
    o = o < 0.0 ? 0.0 : o;                  // torch.clip: max with 0, then min with 255 (NaN propagates through both selects)
    o = o > 255.0 ? 255.0 : o;
    return o;
}

__global__ __launch_bounds__(256) void decode_kernel(const float* __restrict__ pred, double* __restrict__ out, int res_h, int res_w, int P) {
    const int x = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
    if (x >= res_w) return;
    double* d = out + ((size_t)r * res_w + x) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) d[c] = decoded(pred, r, x, c, res_h, res_w / P, P);
}

__global__ __launch_bounds__(256) void mask_kernel(const float* __restrict__ pred, uint8_t* __restrict__ out, int res_h, int res_w, int P) {
    const int x = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
    if (x >= res_w) return;
    double m = decoded(pred, r, x, 0, res_h, res_w / P, P);
    m = m + decoded(pred, r, x, 1, res_h, res_w / P, P);
    m = m + decoded(pred, r, x, 2, res_h, res_w / P, P);
    m = m / 3.0;
    const uint8_t v = m > 128.0 ? 1 : 0;
    uint8_t* d = out + ((size_t)r * res_w + x) * 3;
    d[0] = v; d[1] = v; d[2] = v;
}

// grid: x = blocks of 256 output columns, y = output rows of the full-size frame.  Each frame byte is read once and written once.
__global__ __launch_bounds__(256) void blend_kernel(const float* __restrict__ pred, const uint8_t* __restrict__ image, uint8_t* __restrict__ out,
                                                    int out_w, const int* __restrict__ ytab, const int* __restrict__ xtab, int res_h, int res_w, int P) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= out_w) return;
    const int sr = ytab[y], sx = xtab[x];
    const size_t px = ((size_t)y * out_w + x) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double f = 0.6 * decoded(pred, sr, sx, c, res_h, res_w / P, P);
        f = f / 255.0;
        f = f + 0.4;
        const double v = (double)image[px + c] * f;
        out[px + c] = (uint8_t)(int)v;       // numpy astype(uint8) of a value in [0, 255]: truncation
    }
}

inline dim3 grid2(int w, int h, int z = 1) { return dim3((unsigned)((w + 255) / 256), (unsigned)h, (unsigned)z); }

}  // namespace

extern "C" {

// src rows are src_row_bytes apart (>= src_w * channels): a crop box is a pointer offset plus the parent's row pitch.
int pa_resample_u8_box(const void* src, int64_t src_row_bytes, int src_h, int src_w, void* dst, int dst_h, int dst_w, int channels,
                       const void* bounds, const void* coeffs, int ksize, int vertical, hipStream_t stream) {
    if (channels < 1 || channels > MAXC || dst_h < 1 || dst_w < 1 || dst_h > 65535 || src_row_bytes < (int64_t)src_w * channels)
        return (int)hipErrorInvalidValue;
    if (vertical ? dst_w != src_w : dst_h != src_h) return (int)hipErrorInvalidValue;
    const size_t rb = (size_t)src_row_bytes;
    if (vertical)
        PA_LAUNCH(resample_u8_kernel<true>, grid2(dst_w, dst_h), dim3(256), 0, stream, (const uint8_t*)src, rb, (uint8_t*)dst, dst_w, channels,
                  (const int*)bounds, (const int*)coeffs, ksize);
    else if ((size_t)src_w * channels + 8 <= 48 * 1024)          // the row fits LDS: staged kernel, one workgroup per row
        PA_LAUNCH(resample_h_lds_kernel, dim3((unsigned)dst_h), dim3(256), ((size_t)src_w * channels + 8 + 15) & ~(size_t)15, stream,
                  (const uint8_t*)src, rb, src_w, (uint8_t*)dst, dst_w, channels, (const int*)bounds, (const int*)coeffs, ksize);
    else
        PA_LAUNCH(resample_u8_kernel<false>, grid2(dst_w, dst_h), dim3(256), 0, stream, (const uint8_t*)src, rb, (uint8_t*)dst, dst_w, channels,
                  (const int*)bounds, (const int*)coeffs, ksize);
    LAUNCH_CHECK();
}

int pa_resample_u8(const void* src, int src_h, int src_w, void* dst, int dst_h, int dst_w, int channels, const void* bounds,
                   const void* coeffs, int ksize, int vertical, hipStream_t stream) {
    return pa_resample_u8_box(src, (int64_t)src_w * channels, src_h, src_w, dst, dst_h, dst_w, channels, bounds, coeffs, ksize, vertical, stream);
}

int pa_gather_u8_box(const void* src, int64_t src_row_bytes, int src_h, int src_w, void* dst, int dst_h, int dst_w, int channels,
                     const void* ytab, const void* xtab, hipStream_t stream) {
    (void)src_h;
    if (channels < 1 || dst_h < 1 || dst_w < 1 || dst_h > 65535 || src_row_bytes < (int64_t)src_w * channels) return (int)hipErrorInvalidValue;
    PA_LAUNCH(gather_u8_kernel, grid2(dst_w, dst_h), dim3(256), 0, stream, (const uint8_t*)src, (size_t)src_row_bytes, (uint8_t*)dst, dst_w, channels,
              (const int*)ytab, (const int*)xtab);
    LAUNCH_CHECK();
}

int pa_gather_u8(const void* src, int src_h, int src_w, void* dst, int dst_h, int dst_w, int channels, const void* ytab,
                 const void* xtab, hipStream_t stream) {
    return pa_gather_u8_box(src, (int64_t)src_w * channels, src_h, src_w, dst, dst_h, dst_w, channels, ytab, xtab, stream);
}

int pa_seggpt_stitch(const void* prompts, const void* targets, const void* target_div, const void* query, float* imgs,
                     float* tgts, int n_prompts, int res_h, int res_w, hipStream_t stream) {
    if (n_prompts < 1 || n_prompts > 65535 || res_h < 1 || res_w < 1 || 2 * res_h > 65535) return (int)hipErrorInvalidValue;
    PA_LAUNCH(stitch_kernel, grid2(res_w, 2 * res_h, n_prompts), dim3(256), 0, stream, (const uint8_t*)prompts, (const uint8_t*)targets,
              (const double*)target_div, (const uint8_t*)query, imgs, tgts, res_h, res_w);
    LAUNCH_CHECK();
}

static bool canvas_ok(int res_h, int res_w, int patch) {
    return patch >= 1 && res_h >= 1 && res_w >= 1 && res_h % patch == 0 && res_w % patch == 0 && res_h <= 65535;
}

int pa_seggpt_decode(const float* pred, void* out_f64, int res_h, int res_w, int patch, hipStream_t stream) {
    if (!canvas_ok(res_h, res_w, patch)) return (int)hipErrorInvalidValue;
    PA_LAUNCH(decode_kernel, grid2(res_w, res_h), dim3(256), 0, stream, pred, (double*)out_f64, res_h, res_w, patch);
    LAUNCH_CHECK();
}

int pa_seggpt_mask(const float* pred, void* out_u8, int res_h, int res_w, int patch, hipStream_t stream) {
    if (!canvas_ok(res_h, res_w, patch)) return (int)hipErrorInvalidValue;
    PA_LAUNCH(mask_kernel, grid2(res_w, res_h), dim3(256), 0, stream, pred, (uint8_t*)out_u8, res_h, res_w, patch);
    LAUNCH_CHECK();
}

int pa_seggpt_blend(const float* pred, const void* image, void* out, int out_h, int out_w, const void* ytab, const void* xtab,
                    int res_h, int res_w, int patch, hipStream_t stream) {
    if (!canvas_ok(res_h, res_w, patch) || out_h < 1 || out_w < 1 || out_h > 65535) return (int)hipErrorInvalidValue;
    PA_LAUNCH(blend_kernel, grid2(out_w, out_h), dim3(256), 0, stream, pred, (const uint8_t*)image, (uint8_t*)out, out_w, (const int*)ytab,
              (const int*)xtab, res_h, res_w, patch);
    LAUNCH_CHECK();
}

}  // extern "C"
