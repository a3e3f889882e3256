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

__global__ __launch_bounds__(256) void crop_f32_kernel(const float* __restrict__ src, float* __restrict__ dst, const int* __restrict__ boxes, int C, int H, int W,
                                                       int nearest) {
    const int ox = blockIdx.x * 256 + threadIdx.x, oy = blockIdx.y, bc = blockIdx.z;
    if (ox >= W) return;
    const int b = bc / C;
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
              nearest);
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
