// Pixel work of Painter's training input pipeline on the device: what Painter/data/pairdataset.py:106-190 and the
// transform stack of Painter/main_train.py:232-251 (Painter/data/pair_transforms.py) do per sample on the host with PIL and CPU torch,
// from explicit random parameters (crop boxes, jitter order / factors, flip flags) that the host keeps drawing.
//   RandomResizedCrop on the decoded uint8 pictures  -> pa_resample_u8_box / pa_gather_u8_box (csrc/seggpt_io.hip, Pillow-exact)
//   ColorJitter (PIL ImageEnhance + HSV round trip)   -> pa_color_jitter      (this file; integer / float32 / float64 exactly as Pillow)
//   hflip + ToTensor + Normalize + two-pair stitch    -> pa_to_tensor_normalize
//   second RandomResizedCrop on the float canvases    -> pa_resized_crop_f32  (torch upsample_bicubic2d / nearest semantics)
//   `valid` rules                                     -> pa_pair_valid
// All byte / index / elementwise work, HBM- and launch-bound: one thread per pixel, rows on consecutive lanes, batched over the samples
// of a step with per-sample parameter arrays so a step costs a fixed handful of launches.  No MFMA, no LDS tiles (no reuse to capture).
// Compiled with -ffp-contract=off (build.py) like seggpt_io.hip: Pillow's blend is `in1 + alpha * (in2 - in1)` in float32 with two
// roundings and a truncating cast; a fused multiply-add changes bytes.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/painter_hip.h"
#include "common.h"

#pragma clang fp contract(off)

namespace {

typedef unsigned long long u64;

// float32 division with IEEE rounding irrespective of the compiler's fp32-division mode: the double quotient rounded to float is the
// correctly rounded float quotient (53 >= 2 * 24 + 2 bits, so the double rounding is innocuous).
DEVI float fdiv(float a, float b) { return (float)((double)a / (double)b); }

DEVI int gray_l(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }      // Pillow Convert.c rgb2l

// Pillow Blend.c ImagingBlend, one byte.  alpha is a C float.
DEVI int blend_byte(int in1, int in2, float alpha) {
    if (alpha == 0.0f) return in1;
    if (alpha == 1.0f) return in2;
    const float prod = alpha * (float)(in2 - in1);
    const float t = (float)in1 + prod;
    if (alpha >= 0.0f && alpha <= 1.0f) return (int)t & 0xff;
    return t <= 0.0f ? 0 : (t >= 255.0f ? 255 : (int)t);
}

DEVI int clip255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// Pillow Convert.c rgb2hsv_row
DEVI void rgb2hsv(int r, int g, int b, int& uh, int& us, int& uv) {
    const int maxc = max(r, max(g, b)), minc = min(r, min(g, b));
    uv = maxc;
    if (minc == maxc) {
        uh = 0;
        us = 0;
        return;
    }
    const float cr = (float)(maxc - minc);
    const float s = fdiv(cr, (float)maxc);
    const float rc = fdiv((float)(maxc - r), cr), gc = fdiv((float)(maxc - g), cr), bc = fdiv((float)(maxc - b), cr);
    float h;
    if (r == maxc) h = bc - gc;
    else if (g == maxc) h = (float)((2.0 + (double)rc) - (double)bc);
    else h = (float)((4.0 + (double)gc) - (double)rc);
    const double x = (double)h / 6.0 + 1.0;               // in (0.5, 2): fmod(x, 1) = x - floor(x), exact
    h = (float)(x - floor(x));
    uh = clip255((int)((double)h * 255.0));
    us = clip255((int)((double)s * 255.0));
}

// Pillow Convert.c hsv2rgb
DEVI void hsv2rgb(int h, int s, int v, int& r, int& g, int& b) {
    if (s == 0) {
        r = g = b = v;
        return;
    }
    const double hh = (double)h * 6.0 / 255.0;
    const int i = (int)floor(hh);
    const double f = (double)(float)(hh - (double)i);
    const double fs = (double)(float)((double)s / 255.0);
    const double vd = (double)v;
    const int p = clip255((int)floor(vd * (1.0 - fs) + 0.5));
    const int q = clip255((int)floor(vd * (1.0 - fs * f) + 0.5));
    const int t = clip255((int)floor(vd * (1.0 - fs * (1.0 - f)) + 0.5));
    switch (i % 6) {
        case 0: r = v; g = t; b = p; break;
        case 1: r = q; g = v; b = p; break;
        case 2: r = p; g = v; b = t; break;
        case 3: r = p; g = q; b = v; break;
        case 4: r = t; g = p; b = v; break;
        default: r = v; g = p; b = q; break;
    }
}

// ---- ColorJitter.  ops: int32 [B][4] (0 brightness, 1 contrast, 2 saturation, 3 hue, < 0 nothing), factors: float [B][4] (slot of a
// hue op: the uint8 added to H).  One (sum, apply) kernel pair per slot, every sample doing ITS op of that slot.
__global__ __launch_bounds__(256) void jitter_sum_kernel(const uint8_t* __restrict__ images, const int* __restrict__ ops, int slot,
                                                         u64* __restrict__ sums, int npix) {
    const int b = blockIdx.y;
    if (ops[b * 4 + slot] != 1) return;                    // only contrast needs the mean grey level
    __shared__ unsigned int part;
    if (threadIdx.x == 0) part = 0;
    __syncthreads();
    const uint8_t* img = images + (size_t)b * npix * 3;
    unsigned int local = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < npix; i += gridDim.x * 256) local += (unsigned)gray_l(img[3 * i], img[3 * i + 1], img[3 * i + 2]);
    atomicAdd(&part, local);                               // integer sums: order-independent, deterministic
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&sums[b], (u64)part);
}

__global__ __launch_bounds__(256) void jitter_apply_kernel(uint8_t* __restrict__ images, const int* __restrict__ ops, const float* __restrict__ factors,
                                                           int slot, const u64* __restrict__ sums, int npix) {
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const int op = ops[b * 4 + slot];
    if (op < 0 || op > 3 || i >= npix) return;
    const float f = factors[b * 4 + slot];
    uint8_t* px = images + ((size_t)b * npix + i) * 3;
    const int r = px[0], g = px[1], bl = px[2];
    int o0, o1, o2;
    if (op == 0) {                                         // ImageEnhance.Brightness: blend with black
        o0 = blend_byte(0, r, f); o1 = blend_byte(0, g, f); o2 = blend_byte(0, bl, f);
    } else if (op == 1) {                                  // ImageEnhance.Contrast: blend with int(mean(L) + 0.5)
        const int mean = (int)((double)sums[b] / (double)npix + 0.5);
        o0 = blend_byte(mean, r, f); o1 = blend_byte(mean, g, f); o2 = blend_byte(mean, bl, f);
    } else if (op == 2) {                                  // ImageEnhance.Color: blend with the grey picture
        const int l = gray_l(r, g, bl);
        o0 = blend_byte(l, r, f); o1 = blend_byte(l, g, f); o2 = blend_byte(l, bl, f);
    } else {                                               // torchvision adjust_hue: H += shift (mod 256) in Pillow's HSV
        int h, s, v;
        rgb2hsv(r, g, bl, h, s, v);
        h = (h + (int)f) & 0xff;
        hsv2rgb(h, s, v, o0, o1, o2);
    }
    px[0] = (uint8_t)o0; px[1] = (uint8_t)o1; px[2] = (uint8_t)o2;
}

// ---- hflip + ToTensor + Normalize, written into rows [row0, row0 + h) of a float32 [B][3][canvas_h][w] canvas (two-pair stitch).
__device__ __constant__ float kMeanF[3] = {0.485f, 0.456f, 0.406f};
__device__ __constant__ float kStdF[3] = {0.229f, 0.224f, 0.225f};

__global__ __launch_bounds__(256) void to_tensor_kernel(const uint8_t* __restrict__ images, const int* __restrict__ flip, float* __restrict__ canvas,
                                                        int h, int w, int canvas_h, int row0) {
    const int x = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y, b = blockIdx.z;
    if (x >= w) return;
    const int sx = flip[b] ? w - 1 - x : x;
    const uint8_t* px = images + (((size_t)b * h + r) * w + sx) * 3;
    const size_t plane = (size_t)canvas_h * w;
    float* o = canvas + (size_t)b * 3 * plane + (size_t)(row0 + r) * w + x;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float v = fdiv((float)px[c], 255.0f);
        v = v - kMeanF[c];
        o[c * plane] = fdiv(v, kStdF[c]);
    }
}

// ---- RandomResizedCrop on float32 [B][C][H][W] canvases back to H x W: torch interpolate on the slice (aten UpSample.h).
DEVI float cubic1(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
DEVI float cubic2(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }

// modes: per-sample int32 (0 bicubic, 1 nearest, 2 plain copy of the whole plane) or NULL (then `nearest` holds for every sample)
__global__ __launch_bounds__(256) void crop_f32_kernel(const float* __restrict__ src, float* __restrict__ dst, const int* __restrict__ boxes, int C, int H, int W,
                                                       int nearest, const int* __restrict__ modes) {
    const int ox = blockIdx.x * 256 + threadIdx.x, oy = blockIdx.y, bc = blockIdx.z;
    if (ox >= W) return;
    const int b = bc / C;
    if (modes != nullptr) {
        nearest = modes[b];
        if (nearest == 2) {
            const size_t at = (size_t)bc * H * W + (size_t)oy * W + ox;
            dst[at] = src[at];
            return;
        }
    }
    const int top = boxes[4 * b], left = boxes[4 * b + 1], bh = boxes[4 * b + 2], bw = boxes[4 * b + 3];
    const float* s = src + (size_t)bc * H * W;
    float* d = dst + (size_t)bc * H * W + (size_t)oy * W + ox;
    const float sy = (float)bh / (float)H, sx = (float)bw / (float)W;
    if (nearest) {                                         // nearest_idx: identity / >> 1 shortcuts, else min(floorf(dst * scale), in - 1)
        const int iy = bh == H ? oy : (H == 2 * bh ? oy >> 1 : min((int)floorf((float)oy * sy), bh - 1));
        const int ix = bw == W ? ox : (W == 2 * bw ? ox >> 1 : min((int)floorf((float)ox * sx), bw - 1));
        *d = s[(size_t)(top + iy) * W + left + ix];
        return;
    }
    const float A = -0.75f;
    const float ry = sy * ((float)oy + 0.5f) - 0.5f, rx = sx * ((float)ox + 0.5f) - 0.5f;
    const float fy = floorf(ry), fx = floorf(rx);
    const int iy = (int)fy, ix = (int)fx;
    const float ty = ry - fy, tx = rx - fx;
    const float wy[4] = {cubic2(ty + 1.f, A), cubic1(ty, A), cubic1(1.f - ty, A), cubic2(2.f - ty, A)};
    const float wx[4] = {cubic2(tx + 1.f, A), cubic1(tx, A), cubic1(1.f - tx, A), cubic2(2.f - tx, A)};
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int yy = top + min(max(iy - 1 + j, 0), bh - 1);
        float rowv = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) rowv += s[(size_t)yy * W + left + min(max(ix - 1 + k, 0), bw - 1)] * wx[k];
        acc += rowv * wy[j];
    }
    *d = acc;
}

// ---- RandomResizedCrop for every picture of a step in two launches (pa_resized_crop_u8_batch).  A job = one decoded picture's crop
// box -> one [out_h][out_w][3] output.  Pillow order: horizontal pass into `mid` (all h rows of the box; a plain copy when the width
// does not change, as Pillow skips the pass), then the vertical pass (a copy when the height does not change).  PIL-nearest jobs are
// one gather in the second launch.  The arithmetic is resample_h_lds_kernel's / resample_u8_kernel<true>'s / gather_u8_kernel's
// (csrc/seggpt_io.hip), so the bytes are the per-picture entry points' bytes.
constexpr int CROP_PRECISION_BITS = 32 - 8 - 2;          // Pillow Resample.c
DEVI uint8_t crop_clip8(int acc) {
    const int v = acc >> CROP_PRECISION_BITS;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}
__global__ __launch_bounds__(256) void crop_u8_h_kernel(const pa_crop_job* __restrict__ jobs, uint8_t* __restrict__ mid, int out_w) {
    extern __shared__ __attribute__((aligned(16))) unsigned char row_lds[];
    const pa_crop_job jb = jobs[blockIdx.y];
    const int y = blockIdx.x, tid = threadIdx.x;
    if (y >= jb.h || jb.nearest) return;
    const uint8_t* row = (const uint8_t*)jb.src + (size_t)y * jb.src_row_bytes;
    uint8_t* drow = mid + ((size_t)jb.mid_row0 + y) * out_w * 3;
    const int nbytes = jb.w * 3;
    if (jb.w == out_w) {
        for (int i = tid; i < nbytes; i += 256) drow[i] = row[i];
        return;
    }
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
    const int* bounds = (const int*)jb.xbounds;
    const int* coeffs = (const int*)jb.xcoeffs;
    for (int x = tid; x < out_w; x += 256) {
        const int first = bounds[2 * x], taps = bounds[2 * x + 1];
        const int* k = coeffs + (size_t)x * jb.xksize;
        int a0 = 1 << (CROP_PRECISION_BITS - 1), a1 = a0, a2 = a0;
        const unsigned char* sp = r + first * 3;
        for (int t = 0; t < taps; ++t, sp += 3) {
            const int kt = k[t];
            a0 += (int)sp[0] * kt;
            a1 += (int)sp[1] * kt;
            a2 += (int)sp[2] * kt;
        }
        drow[3 * x] = crop_clip8(a0);
        drow[3 * x + 1] = crop_clip8(a1);
        drow[3 * x + 2] = crop_clip8(a2);
    }
}
__global__ __launch_bounds__(256) void crop_u8_v_kernel(const pa_crop_job* __restrict__ jobs, const uint8_t* __restrict__ mid, int out_h, int out_w) {
    const pa_crop_job jb = jobs[blockIdx.z];
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= out_w) return;
    uint8_t* d = (uint8_t*)jb.dst + ((size_t)y * out_w + x) * 3;
    if (jb.nearest) {
        int sy = y, sx = x;
        if (jb.h != out_h || jb.w != out_w) {               // PIL returns a copy when the size does not change
            sy = ((const int*)jb.ybounds)[y];
            sx = ((const int*)jb.xbounds)[x];
        }
        if (sy < 0 || sx < 0) {
            d[0] = d[1] = d[2] = 0;
            return;
        }
        const uint8_t* sp = (const uint8_t*)jb.src + (size_t)sy * jb.src_row_bytes + (size_t)sx * 3;
        d[0] = sp[0];
        d[1] = sp[1];
        d[2] = sp[2];
        return;
    }
    const size_t mrow = (size_t)out_w * 3;
    const uint8_t* m0 = mid + (size_t)jb.mid_row0 * mrow + (size_t)x * 3;
    if (jb.h == out_h) {
        const uint8_t* sp = m0 + (size_t)y * mrow;
        d[0] = sp[0];
        d[1] = sp[1];
        d[2] = sp[2];
        return;
    }
    const int* bounds = (const int*)jb.ybounds;
    const int first = bounds[2 * y], taps = bounds[2 * y + 1];
    const int* k = (const int*)jb.ycoeffs + (size_t)y * jb.yksize;
    int a0 = 1 << (CROP_PRECISION_BITS - 1), a1 = a0, a2 = a0;
    const uint8_t* sp = m0 + (size_t)first * mrow;
    for (int t = 0; t < taps; ++t, sp += mrow) {
        const int kt = k[t];
        a0 += (int)sp[0] * kt;
        a1 += (int)sp[1] * kt;
        a2 += (int)sp[2] * kt;
    }
    d[0] = crop_clip8(a0);
    d[1] = crop_clip8(a1);
    d[2] = crop_clip8(a2);
}

// ---- `valid` rules (pairdataset.py:152-180).  modes: 0 ones, 1 (target < thres -> 0), 2 pose (target > thres -> 10, fewer than 300
// foreground elements -> all 0), 3 (fewer than 300 foreground elements -> all 0).
__global__ __launch_bounds__(256) void valid_count_kernel(const float* __restrict__ tgts, const int* __restrict__ modes, const float* __restrict__ thres,
                                                          int* __restrict__ counts, int plane) {
    const int b = blockIdx.z, c = blockIdx.y;
    if (modes[b] < 2) return;
    __shared__ int part;
    if (threadIdx.x == 0) part = 0;
    __syncthreads();
    const float* t = tgts + ((size_t)b * 3 + c) * plane;
    const float th = thres[b * 3 + c];
    int local = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < plane; i += gridDim.x * 256) local += t[i] > th ? 1 : 0;
    atomicAdd(&part, local);
    __syncthreads();
    if (threadIdx.x == 0 && part) atomicAdd(&counts[b], part);
}

__global__ __launch_bounds__(256) void valid_apply_kernel(const float* __restrict__ tgts, float* __restrict__ valid, const int* __restrict__ modes,
                                                          const float* __restrict__ thres, const int* __restrict__ counts, int plane) {
    const int b = blockIdx.z, c = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= plane) return;
    const size_t o = ((size_t)b * 3 + c) * plane + i;
    const int mode = modes[b];
    float v = 1.0f;
    if (mode == 1) v = tgts[o] < thres[b * 3 + c] ? 0.0f : 1.0f;
    else if (mode == 2) v = counts[b] < 300 ? 0.0f : (tgts[o] > thres[b * 3 + c] ? 10.0f : 1.0f);
    else if (mode == 3) v = counts[b] < 300 ? 0.0f : 1.0f;
    valid[o] = v;
}

inline unsigned blocks(int n) { return (unsigned)((n + 255) / 256); }

}  // namespace

extern "C" {

int64_t pa_color_jitter_workspace_bytes(int batch) { return (int64_t)4 * (batch > 0 ? batch : 1) * (int64_t)sizeof(u64); }

// ops_host: HOST copy of ops (int32 [batch][4]) or NULL -- lets the launcher skip the slots nobody uses and the mean pass when no
// sample has a contrast op in that slot.
int pa_color_jitter(void* images, const void* ops, const void* factors, const void* ops_host, void* workspace, int batch, int h, int w,
                    hipStream_t stream) {
    if (batch < 1 || batch > 65535 || h < 1 || w < 1 || (int64_t)h * w > (1 << 24)) return (int)hipErrorInvalidValue;      // 32-bit grey sums per block
    const int npix = h * w;
    u64* sums = (u64*)workspace;
    hipError_t e = hipMemsetAsync(sums, 0, (size_t)pa_color_jitter_workspace_bytes(batch), stream);
    if (e != hipSuccess) return (int)e;
    const int* oh = (const int*)ops_host;
    for (int slot = 0; slot < 4; ++slot) {
        bool any = oh == nullptr, contrast = oh == nullptr;
        for (int b = 0; oh && b < batch; ++b) {
            any |= oh[b * 4 + slot] >= 0;
            contrast |= oh[b * 4 + slot] == 1;
        }
        if (!any) continue;
        if (contrast)
            PA_LAUNCH(jitter_sum_kernel, dim3(min(blocks(npix), 64u), (unsigned)batch), dim3(256), 0, stream, (const uint8_t*)images, (const int*)ops, slot,
                      sums + (size_t)slot * batch, npix);
        PA_LAUNCH(jitter_apply_kernel, dim3(blocks(npix), (unsigned)batch), dim3(256), 0, stream, (uint8_t*)images, (const int*)ops, (const float*)factors,
                  slot, sums + (size_t)slot * batch, npix);
    }
    LAUNCH_CHECK();
}

int pa_to_tensor_normalize(const void* images, const void* flip, float* canvas, int batch, int h, int w, int canvas_h, int row0,
                           hipStream_t stream) {
    if (batch < 1 || batch > 65535 || h < 1 || h > 65535 || w < 1 || row0 < 0 || row0 + h > canvas_h) return (int)hipErrorInvalidValue;
    PA_LAUNCH(to_tensor_kernel, dim3(blocks(w), (unsigned)h, (unsigned)batch), dim3(256), 0, stream, (const uint8_t*)images, (const int*)flip, canvas, h, w,
              canvas_h, row0);
    LAUNCH_CHECK();
}

int pa_resized_crop_f32(const float* src, float* dst, const void* boxes, int batch, int channels, int h, int w, int nearest,
                        hipStream_t stream) {
    if (batch < 1 || channels < 1 || (int64_t)batch * channels > 65535 || h < 1 || h > 65535 || w < 1 || src == dst) return (int)hipErrorInvalidValue;
    PA_LAUNCH(crop_f32_kernel, dim3(blocks(w), (unsigned)h, (unsigned)(batch * channels)), dim3(256), 0, stream, src, dst, (const int*)boxes, channels, h, w,
              nearest, (const int*)nullptr);
    LAUNCH_CHECK();
}

int pa_resized_crop_f32_modes(const float* src, float* dst, const void* boxes, const void* modes, int batch, int channels, int h, int w,
                              hipStream_t stream) {
    if (batch < 1 || channels < 1 || (int64_t)batch * channels > 65535 || h < 1 || h > 65535 || w < 1 || src == dst || modes == nullptr)
        return (int)hipErrorInvalidValue;
    PA_LAUNCH(crop_f32_kernel, dim3(blocks(w), (unsigned)h, (unsigned)(batch * channels)), dim3(256), 0, stream, src, dst, (const int*)boxes, channels, h, w,
              0, (const int*)modes);
    LAUNCH_CHECK();
}

int pa_resized_crop_u8_batch(const pa_crop_job* jobs, void* mid, int n_jobs, int max_h, int max_w, int out_h, int out_w, hipStream_t stream) {
    if (n_jobs < 1 || n_jobs > 65535 || max_h < 1 || max_w < 1 || out_h < 1 || out_h > 65535 || out_w < 1) return (int)hipErrorInvalidValue;
    const size_t lds = ((size_t)max_w * 3 + 8 + 15) & ~(size_t)15;
    if (lds > 64 * 1024) return (int)hipErrorInvalidValue;          // a source row is staged in LDS
    PA_LAUNCH(crop_u8_h_kernel, dim3((unsigned)max_h, (unsigned)n_jobs), dim3(256), lds, stream, jobs, (uint8_t*)mid, out_w);
    PA_LAUNCH(crop_u8_v_kernel, dim3(blocks(out_w), (unsigned)out_h, (unsigned)n_jobs), dim3(256), 0, stream, jobs, (const uint8_t*)mid, out_h, out_w);
    LAUNCH_CHECK();
}

int64_t pa_pair_valid_workspace_bytes(int batch) { return (int64_t)(batch > 0 ? batch : 1) * (int64_t)sizeof(int); }

int pa_pair_valid(const float* tgts, float* valid, const void* modes, const void* thres, void* workspace, int batch, int plane,
                  hipStream_t stream) {
    if (batch < 1 || batch > 65535 || plane < 1) return (int)hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(workspace, 0, (size_t)pa_pair_valid_workspace_bytes(batch), stream);
    if (e != hipSuccess) return (int)e;
    PA_LAUNCH(valid_count_kernel, dim3(min(blocks(plane), 64u), 3u, (unsigned)batch), dim3(256), 0, stream, tgts, (const int*)modes, (const float*)thres,
              (int*)workspace, plane);
    PA_LAUNCH(valid_apply_kernel, dim3(blocks(plane), 3u, (unsigned)batch), dim3(256), 0, stream, tgts, valid, (const int*)modes, (const float*)thres,
              (const int*)workspace, plane);
    LAUNCH_CHECK();
}

}  // extern "C"
