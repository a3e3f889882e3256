/* painter_hip.h -- C ABI of libpainter_hip.so: the MI355X (gfx950) kernels behind the Painter / SegGPT
 * ViT forward/backward hot path.
 *
 * The reference (baaivision/Painter) has no FFI of its own: its hot path is a chain of ATen ops issued from
 * Python.  Each entry point below is the fused replacement for one group of those ops;
 * the comment on each cites the reference lines it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - plain pointers + sizes, no torch types; every pointer is a DEVICE pointer unless noted.
 *   - `dtype` selects the operand/activation storage type T: PA_F32 (exact fp32 MFMA, parity build) or
 *     PA_BF16 (bf16 operands, fp32 accumulate).  Parameters, the residual stream, statistics, losses and all
 *     parameter gradients are fp32 in both builds.  "T*" in a comment means float* or bf16* per `dtype`.
 *   - every function only ENQUEUES work on `stream` (no allocation, no synchronisation) and returns a
 *     hipError_t as int (0 = success).  Buffers are borrowed for the duration of the enqueued work.
 *   - `ld*` arguments are row strides in ELEMENTS.  16-byte alignment of all base pointers is required.
 *   - *_workspace_bytes() are host-only helpers giving the scratch size the matching call needs.
 */
#ifndef PAINTER_HIP_H
#define PAINTER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef __HIP__
typedef struct ihipStream_t* hipStream_t;
#endif

enum { PA_F32 = 0, PA_BF16 = 1 };
enum {
    PA_EPI_BIAS = 0,       /* out(T)   = x.W^T + b                                   */
    PA_EPI_BIAS_F32 = 1,   /* out(f32) = x.W^T + b                                   */
    PA_EPI_BIAS_GELU = 2,  /* out2(T) = x.W^T + b (pre-activation, may be NULL); out(T) = gelu_erf(out2) */
    PA_EPI_BIAS_RESID = 3  /* out(f32) = resid + rowscale[row / rows_per_sample] * (x.W^T + b) */
};

int pa_abi_version(void);

/* ---- nn.Linear: y = x W^T + b.  models_painter.py:76 (qkv), :87 (proj), timm Mlp fc1/fc2 (:201,:230) ---- */
int pa_linear_fwd(int dtype, int epilogue, const void* x /*T [M,K]*/, int64_t ldx, const void* w /*T [N,K]*/,
                  const float* bias /*[N]*/, void* out, void* out2, int64_t ldo, const float* resid /*f32 [M,N] ld=ldo*/,
                  const float* rowscale /*[M/rows_per_sample] or NULL*/, int rows_per_sample, int M, int N, int K,
                  hipStream_t stream);
/* decoder_embed + pixel shuffle 'nhwpqc->nchpwq' (models_painter.py:423-428); output is NHWC [B, Hp*P, Wp*P, C] T */
int pa_linear_pixshuf(int dtype, const void* x, int64_t ldx, const void* w /*T [P*P*C, K]*/, const float* bias,
                      void* out_nhwc, int batch, int Hp, int Wp, int P, int C, int K, hipStream_t stream);
/* autograd of the above: dX = dY.W (optionally * gelu'(pre)), dW = dY^T.X (fp32), db = colsum(dY) */
int pa_linear_dgrad(int dtype, const void* dy /*T [M,N]*/, int64_t lddy, const void* w /*T [N,K]*/,
                    const void* pre_for_dgelu /*T [M,K] ld=lddx or NULL*/, void* dx /*T [M,K]*/, int64_t lddx, int M, int N,
                    int K, hipStream_t stream);
int64_t pa_linear_wgrad_workspace_bytes(int dtype, int M, int N, int K);
int pa_linear_wgrad(int dtype, const void* dy /*T [M,N]*/, int64_t lddy, const void* x /*T [M,K]*/, int64_t ldx,
                    float* dw /*f32 [N,K]*/, void* workspace, int M, int N, int K, hipStream_t stream);
int64_t pa_colsum_workspace_bytes(int M, int N);
int pa_colsum(int dtype, const void* x /*T [M,N]*/, int64_t ld, int M, int N, float* out /*[N]*/, void* workspace,
              hipStream_t stream);
int pa_slab_reduce(const float* in, float* out, int64_t n, int nz, int64_t stride, int accumulate, hipStream_t stream);

/* ---- nn.LayerNorm(eps=1e-6) over channels: models_painter.py:218,230 (norm1/2), :416-417 (shared tap norm) ---- */
int pa_layernorm_fwd(int dtype, const float* x, int64_t ldx, const float* gamma, const float* beta, float eps,
                     void* y /*T, may be a column slice of the tap concat buffer*/, int64_t ldy, float* mean, float* rstd,
                     int R, int D, hipStream_t stream);
int64_t pa_layernorm_bwd_workspace_bytes(int R, int D);
/* dx = (dres ? dres : 0) + LN'(dy); dres may alias dx.  dxT (optional, T) = rowscale[row/rows_per_sample] * dx.
 * dgamma_dbeta: f32 [2, D], overwritten. */
int pa_layernorm_bwd(int dtype, const void* dy, int64_t lddy, const float* x, int64_t ldx, const float* mean,
                     const float* rstd, const float* gamma, const float* dres, float* dx, int64_t lddx, void* dxT,
                     int64_t lddxT, const float* rowscale, int rows_per_sample, float* dgamma_dbeta, void* workspace,
                     int R, int D, hipStream_t stream);

/* ---- Attention with decomposed rel-pos bias: models_painter.py:76-86 + util/vitdet_utils.py:63-125 ---- */
int pa_relpos_rows_padded(int Hp, int Wp);
/* rcat: T [pa_relpos_rows_padded, 64] = [rel_pos_h ; rel_pos_w ; 0] */
int pa_relpos_pack(int dtype, const float* rel_pos_h, const float* rel_pos_w, void* rcat, int Hp, int Wp,
                   hipStream_t stream);
/* qkv: T [batch*L, 3*heads*64] as produced by the qkv Linear; out: T [batch*L, heads*64]; lse: f32 [batch*heads, L] */
int pa_attn_fwd(int dtype, const void* qkv, int64_t ldq, const void* rcat, void* out, int64_t ldo, float* lse, int batch,
                int L, int heads, int Hp, int Wp, float scale, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PAINTER_HIP_H */
