// Fused optimizer step for the hot path's 296 fp32 parameter tensors (370.7 M elements) -- SURVEY.md 8(f) N1:
//   GradScaler.unscale_ + clip_grad_norm_(max_norm) + AdamW over the layer-decay parameter groups, as issued by
//   Painter/util/misc.py:256-268 (NativeScalerWithGradNormCount.__call__) on the optimizer built at
//   Painter/main_train.py:344-348 (torch.optim.AdamW over util/lr_decay.py:15-61 groups).
// The reference runs this as ~10 multi-tensor ATen passes over parameters/gradients/moments; here it is two:
//   pa_grad_sumsq   one read of every gradient: per-chunk sums of squares (+ a non-finite count), reduced in a fixed order
//   pa_adamw_step   one read-modify-write of p, m, v: unscale, clip coefficient and the inf/nan skip are applied on the fly
//                   from the device-side result of pass 1 -- no host synchronisation anywhere.
// Pass 2 can also refresh a bf16 copy of the parameter (the GEMM operand cache of the engine) in the same sweep, which removes the
// per-step fp32 -> bf16 cast pass over the weight matrices (2.2 GB of traffic).
// Both are pure HBM streaming kernels (16-byte accesses, one 256-thread workgroup per 8192-element chunk; a chunk never
// straddles two tensors).  Algorithmic bytes: pass 1 = 4 B/element, pass 2 = 28 B/element (read p, g, m, v; write p, m, v).
#include "common.h"
#include "../../include/painter_hip.h"

namespace {

constexpr int OPT_CHUNK = 8192;          // elements per workgroup
constexpr int OPT_NT = 256;

struct TensorRec {                        // one per parameter tensor (device table, PaOptTensor in the header)
    float* p;
    const float* g;
    float* m;
    float* v;
    bf16* w16;                            // optional bf16 copy of the parameter kept in step with it (GEMM operand cache), or NULL
    int64_t n;
    int32_t group;
    int32_t first_chunk;                  // index of this tensor's first chunk in the flat chunk order
};
static_assert(sizeof(TensorRec) == sizeof(PaOptTensor), "header / kernel table layout drift");

// tensor owning flat chunk c: last t with first_chunk[t] <= c   (ntensors <= a few hundred: 9 scalar steps)
DEVI int find_tensor(const TensorRec* tab, int nt, int c) {
    int lo = 0, hi = nt - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab[mid].first_chunk <= c) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__global__ __launch_bounds__(OPT_NT) void grad_sumsq_kernel(const TensorRec* __restrict__ tab, int nt, float* __restrict__ part) {
    const int c = blockIdx.x, t = find_tensor(tab, nt, c);
    const TensorRec r = tab[t];
    const int64_t base = (int64_t)(c - r.first_chunk) * OPT_CHUNK;
    const int64_t end = min(r.n, base + OPT_CHUNK);
    float s = 0.f, bad = 0.f;
    if (r.g != nullptr) {
        const float* g = r.g + base;
        const int len = (int)(end - base);
        if ((reinterpret_cast<uintptr_t>(g) & 15) == 0) {
            for (int i = threadIdx.x * 4; i + 3 < len; i += OPT_NT * 4) {
                const float4 x = *reinterpret_cast<const float4*>(g + i);
                s += (x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w);
            }
            for (int i = (len & ~3) + threadIdx.x; i < len; i += OPT_NT) s += g[i] * g[i];
        } else {
            for (int i = threadIdx.x; i < len; i += OPT_NT) s += g[i] * g[i];
        }
    }
    if (!(fabsf(s) <= 3.0e38f)) { bad = 1.f; s = 0.f; }      // inf or nan in this thread's share
    __shared__ float red[2][OPT_NT / 64];
    s = wave_sum(s);
    bad = wave_sum(bad);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = bad; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[2 * (size_t)c] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        part[2 * (size_t)c + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

// out[0] = sum of squares (double accumulation, fixed order), out[1] = number of chunks with a non-finite share
__global__ __launch_bounds__(256) void grad_sumsq_final_kernel(const float* __restrict__ part, int nchunks, float* __restrict__ out) {
    double s = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < nchunks; i += 256) { s += (double)part[2 * (size_t)i]; b += (double)part[2 * (size_t)i + 1]; }
    __shared__ double rs[256], rb[256];
    rs[threadIdx.x] = s; rb[threadIdx.x] = b;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) { rs[threadIdx.x] += rs[threadIdx.x + o]; rb[threadIdx.x] += rb[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = (float)rs[0]; out[1] = (float)rb[0]; }
}

// torch.optim.AdamW single-tensor update, in its operation order (torch/optim/adamw.py, _single_tensor_adamw):
//   p *= 1 - lr*wd;  m += (g - m)*(1 - b1);  v = v*b2 + (1 - b2)*g*g;  p -= (lr / bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
DEVI void adamw1(float& p, float& m, float& v, float g, float lr, float wd, float b1, float b2, float eps, float step_size, float rsqrt_bc2) {
    p = p * (1.f - lr * wd);
    m = m + (g - m) * (1.f - b1);
    v = v * b2 + (1.f - b2) * g * g;
    const float denom = sqrtf(v) * rsqrt_bc2 + eps;
    p = p - step_size * (m / denom);
}

// norm_info[0] is the sum of squares of the gradients AS STORED (still loss-scaled): with a large loss scale it can overflow fp32
// although no single element did.  torch unscales before it takes the norm and has no such window; here an overflowed (or NaN) sum
// counts as "found inf": the step is skipped and GradScaler backs the scale off, instead of clipping every gradient to zero.
DEVI bool skip_step(const float* norm_info, const float* found_inf) {
    return (found_inf != nullptr && found_inf[0] != 0.f) || (norm_info != nullptr && (norm_info[1] != 0.f || !(norm_info[0] <= 3.0e38f)));
}
// steps[g] += 1 for every group that takes part in this step, unless the step is skipped (inf/nan): the per-group step count
// lives on the device so that the skip needs no host round trip (torch's capturable / fused AdamW keeps it there too)
__global__ void opt_advance_kernel(float* __restrict__ steps, PaOptGroups grp, const float* __restrict__ norm_info,
                                   const float* __restrict__ found_inf) {
    const int g = threadIdx.x;
    if (g < 64 && grp.active[g] && !skip_step(norm_info, found_inf)) steps[g] += 1.f;
}

__global__ __launch_bounds__(OPT_NT) void adamw_step_kernel(const TensorRec* __restrict__ tab, int nt, PaOptGroups grp, float b1, float b2, float eps,
                                                            const float* __restrict__ steps, const float* __restrict__ norm_info,
                                                            const float* __restrict__ scale_ptr, const float* __restrict__ found_inf,
                                                            float max_norm) {
    // device-side scalars: skip on inf/nan (GradScaler semantics), unscale factor, clip coefficient
    if (skip_step(norm_info, found_inf)) return;
    float gmul = 1.f;
    if (scale_ptr != nullptr) gmul = 1.f / scale_ptr[0];
    if (norm_info != nullptr) {
        if (max_norm > 0.f) {
            const float total = sqrtf(norm_info[0]) * gmul;                   // norm of the unscaled gradients
            const float coef = max_norm / (total + 1e-6f);                    // torch.nn.utils.clip_grad_norm_
            gmul *= coef < 1.f ? coef : 1.f;
        }
    }
    const int c = blockIdx.x, t = find_tensor(tab, nt, c);
    const TensorRec r = tab[t];
    if (r.g == nullptr) return;
    const int gi = r.group;
    const double stepv = (double)steps[gi];                    // already advanced by opt_advance_kernel
    const float bc1 = (float)(1.0 - pow((double)b1, stepv)), bc2 = (float)(1.0 - pow((double)b2, stepv));
    const float lr = grp.lr[gi], wd = grp.wd[gi], step_size = lr / bc1, rsqrt_bc2 = 1.f / sqrtf(bc2);
    const int64_t base = (int64_t)(c - r.first_chunk) * OPT_CHUNK;
    const int len = (int)(min(r.n, base + OPT_CHUNK) - base);
    float* p = r.p + base;
    float* m = r.m + base;
    float* v = r.v + base;
    const float* g = r.g + base;
    bf16* w16 = r.w16 ? r.w16 + base : nullptr;
    const bool al = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(g) |
                      (reinterpret_cast<uintptr_t>(w16) << 1)) & 15) == 0;
    if (al) {
        for (int i = threadIdx.x * 4; i + 3 < len; i += OPT_NT * 4) {
            float4 pp = *reinterpret_cast<const float4*>(p + i), mm = *reinterpret_cast<const float4*>(m + i),
                   vv = *reinterpret_cast<const float4*>(v + i);
            const float4 gg = *reinterpret_cast<const float4*>(g + i);
            adamw1(pp.x, mm.x, vv.x, gg.x * gmul, lr, wd, b1, b2, eps, step_size, rsqrt_bc2);
            adamw1(pp.y, mm.y, vv.y, gg.y * gmul, lr, wd, b1, b2, eps, step_size, rsqrt_bc2);
            adamw1(pp.z, mm.z, vv.z, gg.z * gmul, lr, wd, b1, b2, eps, step_size, rsqrt_bc2);
            adamw1(pp.w, mm.w, vv.w, gg.w * gmul, lr, wd, b1, b2, eps, step_size, rsqrt_bc2);
            *reinterpret_cast<float4*>(p + i) = pp;
            *reinterpret_cast<float4*>(m + i) = mm;
            *reinterpret_cast<float4*>(v + i) = vv;
            if (w16) *reinterpret_cast<uint2*>(w16 + i) = make_uint2(pack_bf16x2(pp.x, pp.y), pack_bf16x2(pp.z, pp.w));
        }
        for (int i = (len & ~3) + threadIdx.x; i < len; i += OPT_NT) {
            adamw1(p[i], m[i], v[i], g[i] * gmul, lr, wd, b1, b2, eps, step_size, rsqrt_bc2);
            if (w16) w16[i] = (bf16)p[i];
        }
    } else {
        for (int i = threadIdx.x; i < len; i += OPT_NT) {
            adamw1(p[i], m[i], v[i], g[i] * gmul, lr, wd, b1, b2, eps, step_size, rsqrt_bc2);
            if (w16) w16[i] = (bf16)p[i];
        }
    }
}

}   // namespace

extern "C" int pa_opt_chunk_elems(void) { return OPT_CHUNK; }
extern "C" int64_t pa_grad_sumsq_workspace_bytes(int nchunks) { return (int64_t)nchunks * 2 * sizeof(float); }

extern "C" int pa_grad_sumsq(const PaOptTensor* table, int ntensors, int nchunks, float* out2, void* workspace, hipStream_t st) {
    if (ntensors <= 0 || nchunks <= 0) return (int)hipErrorInvalidValue;
    float* part = reinterpret_cast<float*>(workspace);
    PA_LAUNCH(grad_sumsq_kernel, dim3(nchunks), dim3(OPT_NT), 0, st, reinterpret_cast<const TensorRec*>(table), ntensors, part);
    int e = (int)hipGetLastError();
    if (e) return e;
    PA_LAUNCH(grad_sumsq_final_kernel, dim3(1), dim3(256), 0, st, part, nchunks, out2);
    LAUNCH_CHECK();
}

extern "C" int pa_adamw_step(const PaOptTensor* table, int ntensors, int nchunks, const PaOptGroups* groups, float beta1, float beta2, float eps,
                             float* steps, const float* norm_info, const float* grad_scale, const float* found_inf, float max_norm,
                             hipStream_t st) {
    if (ntensors <= 0 || nchunks <= 0 || groups == nullptr || steps == nullptr) return (int)hipErrorInvalidValue;
    PA_LAUNCH(opt_advance_kernel, dim3(1), dim3(64), 0, st, steps, *groups, norm_info, found_inf);
    int e = (int)hipGetLastError();
    if (e) return e;
    PA_LAUNCH(adamw_step_kernel, dim3(nchunks), dim3(OPT_NT), 0, st, reinterpret_cast<const TensorRec*>(table), ntensors, *groups, beta1, beta2, eps,
              steps, norm_info, grad_scale, found_inf, max_norm);
    LAUNCH_CHECK();
}
