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

/* Bumped whenever an entry point's argument list changes (2: `tables` in pa_attn_fwd / pa_attn_bwd; 3: `head_dim` in the attention and
 * rel-pos entry points; 4: `dxT_colsum` in pa_layernorm_bwd, `relpos_part` in pa_attn_bwd, `dx_colsum` in pa_linear_dgrad; 5: the bf16 GELU side output is gelu'(pre), pa_debug_set(9) is a test knob of the conv3x3 weight gradient, pa_attn4_trace is gone, pa_attn_bwd takes `out` / `ldo`, pa_debug_get / pa_attn_launch_counts are new; 6: the bf16 GELU side output is an 8-bit code (uint8, row pitch ldo bytes), pa_debug_set knobs 10 .. 15).  painter_amd/_lib.py refuses a library whose pa_abi_version() differs from the header it parsed. */
#define PA_ABI_VERSION 6
int pa_abi_version(void);
/* diagnostics only (tools/): which = 0 start-up stagger of alternate workgroup rows of the 256x256 GEMM in shader cycles,
 * 1 drop that kernel's epilogue stores (never set by the product path); 2 = tile order of that kernel: 0 blocked 4 x 8 patches per XCD and, for
 * split-K launches, one contiguous (split, tile) run per XCD (default), 1 plain row-major, 2 blocked with split = blockIdx.y (before round 5); 3 = TUNING, set by the engine: target number of workgroups
 * of the weight-gradient GEMM (0 = 256, the whole chip; 64 when the weight gradients run on a side stream); 4 = row-tile height of the un-split bf16
 * GEMMs: 0 by rule (224 rows where that fills the last round of workgroups better), 1 always 256, 2 224 wherever possible; 5 = (G256_ILV_AB experiment
 * builds only) 1 + the DMA-placement schedule of that kernel; 6 = K splits of the
 * rel-pos table-gradient GEMM; 7 = rel-pos table gradient inside the generation-3 dQ kernel: 0 default (on), 1 off, 2 on (tests);
 * 8 = generation-3 attention: workgroups with idle waves dispatched last: 0 default (off since round 5), 1 off, 2 on;
 * 9 = tests: cap on the workgroups of the conv3x3 weight-gradient kernel (0 = 512), so that small images make a workgroup walk many tiles;
 * 10 = LayerNorm backward: 0 rows split over the workgroup's waves wherever D >= 1024 (default), 1 one wave per row everywhere (the kernel of
 * rounds 1 - 4) -- index 5 until round 5, where it collided with the ILV override; 11 .. 15 = round-6 experiment knobs (csrc/common.h). */
int pa_debug_set(int which, int value);
int pa_debug_get(int which);      /* the value last set (-1: no such knob) -- callers that change a knob temporarily restore what they found */

/* ---- nn.Linear: y = x W^T + b.  models_painter.py:76 (qkv), :87 (proj), timm Mlp fc1/fc2 (:201,:230) ----
 * PA_EPI_BIAS_GELU: out = gelu(pre), pre = x W^T + b rounded to T; out2 (optional) receives what the backward needs of pre, to be handed
 * to pa_linear_dgrad as `gelu_aux`:
 *   PA_F32 : pre itself, f32 [M,N] ld = ldo (erf-GELU' is evaluated on it there, to fp32 accuracy);
 *   PA_BF16: the DERIVATIVE gelu'(pre) = Phi(pre) + pre phi(pre) (ABI 5: the forward epilogue has both factors in registers, and the fc2 data
 *            gradient's epilogue becomes one load and one multiply) -- since ABI 6 as an 8-BIT CODE, uint8 [M,N] with a row pitch of ldo BYTES:
 *            q = round((g' + 0.13) * 255 / 1.26), g' = q * 1.26 / 255 - 0.13 (g' lies in [-0.129, 1.129]; absolute error <= 2.5e-3).  Half the
 *            bytes of the bf16 form on the fc1 write and on the fc2 data-gradient read; parity measured unchanged. */
int pa_linear_fwd(int dtype, int epilogue, const void* x /*T [M,K]*/, int64_t ldx, const void* w /*T [N,K]*/,
                  const float* bias /*[N]*/, void* out, void* out2, int64_t ldo, const float* resid /*f32 [M,N] ld=ldo*/,
                  const float* rowscale /*[M/rows_per_sample] or NULL*/, int rows_per_sample, int M, int N, int K,
                  hipStream_t stream);
/* decoder_embed + pixel shuffle 'nhwpqc->nchpwq' (models_painter.py:423-428); output is NHWC [B, Hp*P, Wp*P, C] T */
int pa_linear_pixshuf(int dtype, const void* x, int64_t ldx, const void* w /*T [P*P*C, K]*/, const float* bias,
                      void* out_nhwc, int batch, int Hp, int Wp, int P, int C, int K, hipStream_t stream);
/* autograd of the above: dX = dY.W (optionally * gelu'(pre), from the `out2` of pa_linear_fwd(PA_EPI_BIAS_GELU) of the same dtype), dW = dY^T.X (fp32), db = colsum(dY).
 * dx_colsum (optional, f32 [K], with workspace = pa_linear_dgrad_workspace_bytes(M, K)): the column sums of dX (fp32, in front of its rounding to T on the bf16 fast path) -- dX is the dY
 * of the layer in front (fc1 behind the GELU), so this is that layer's bias gradient, taken in the GEMM's epilogue instead of a second
 * pass over dX. */
int64_t pa_linear_dgrad_workspace_bytes(int M, int K);
int pa_linear_dgrad(int dtype, const void* dy /*T [M,N]*/, int64_t lddy, const void* w /*T [N,K]*/,
                    const void* gelu_aux /*NULL, or what pa_linear_fwd(PA_EPI_BIAS_GELU) put into out2: f32 [M,K] ld=lddx | uint8 [M,K] pitch lddx bytes*/, void* dx /*T [M,K]*/, int64_t lddx, float* dx_colsum,
                    void* workspace, int M, int N, int K, hipStream_t stream);
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
 * dgamma_dbeta: f32 [2, D], overwritten.  dxT_colsum (optional, f32 [D], needs dxT): the column sums of rowscale * dx (the fp32 values
 * dxT is rounded from) -- dxT is the dY of the nn.Linear in front of this norm's residual branch, so this is that layer's bias gradient, fused here. */
int pa_layernorm_bwd(int dtype, const void* dy, int64_t lddy, const float* x, int64_t ldx, const float* mean,
                     const float* rstd, const float* gamma, const float* dres, float* dx, int64_t lddx, void* dxT,
                     int64_t lddxT, const float* rowscale, int rows_per_sample, float* dgamma_dbeta, float* dxT_colsum,
                     void* workspace, int R, int D, hipStream_t stream);
/* Deferred form: pa_layernorm_bwd with dgamma_dbeta = NULL leaves the per-workgroup partial rows in `workspace` (dxT_colsum non-NULL there
 * = "with column sums"; nothing is written through it); pa_layernorm_bwd_reduce finishes them -- nothing downstream in the backward reads
 * these parameter gradients, so the engine runs the reduction on its side stream. */
int pa_layernorm_bwd_reduce(const void* workspace, float* dgamma_dbeta, float* dxT_colsum, int with_colsum, int R, int D,
                            hipStream_t stream);

/* ---- Attention with decomposed rel-pos bias: models_painter.py:76-86 + util/vitdet_utils.py:63-125 ---- */
int pa_relpos_rows_padded(int Hp, int Wp);
/* head_dim (hd) = embed_dim / heads: 64 (every reference factory; all kernel generations) or 80 (ViT-H/14, BASELINE configs[4]: the
 * generic kernels of csrc/attn_fwd.hip / attn_bwd.hip, both dtypes); anything else is rejected.
 * rcat: T [pa_relpos_rows_padded, hd] = [rel_pos_h ; rel_pos_w ; 0] */
int pa_relpos_pack(int dtype, const float* rel_pos_h, const float* rel_pos_w, void* rcat, int Hp, int Wp, int head_dim,
                   hipStream_t stream);
/* Every block's Rcat and Rcat^T (pa_relpos_pack_t below) in ONE launch: tabs = device array of 2 * nblocks `const float*` -- rel_pos_h of
 * every block, then rel_pos_w of every block (util/vitdet_utils.py:63-125 reads these per block); rcat: T [nblocks, NRP, hd]; rcatT: T
 * [nblocks, hd, NRP].  Values identical to the per-block entry points; what a training step calls after the optimizer has written the tables. */
int pa_relpos_pack_batch(int dtype, const void* tabs, void* rcat, void* rcatT, int nblocks, int Hp, int Wp, int head_dim, hipStream_t stream);
/* qkv: T [batch*L, 3*heads*hd] as produced by the qkv Linear; out: T [batch*L, heads*hd]; lse: f32 [batch*heads, L].
 * tables: NULL (inference), or pa_attn_tables_bytes() of device memory that receives the per-query bias tables the
 * backward reuses (only the 28-token-wide bf16 kernels write it; the size is 0 for every other case). */
int64_t pa_attn_tables_bytes(int dtype, int batch, int L, int heads, int Hp, int Wp, int head_dim);
/* 0 = default kernels for the grid (generation 3 where it applies), 3 = the same explicitly, 2 = never use the 28-token-wide
 * generation-3 kernels (diagnostics, A/B, cross-generation tests) */
int pa_attn_set_generation(int generation);
/* diagnostics / tests: host-side launch counts since process start, out6 = {pa_attn_fwd on the generic kernels (csrc/attn_fwd.hip),
 * on generation 2 (attn2.hip: every head_dim-80 grid with key rows of 12..32 tokens), on generation 3 (attn3.hip), pa_attn_bwd likewise} */
int pa_attn_launch_counts(long long* out6);
/* diagnostics: enable != 0 runs the generation-3 dQ kernel with s_memtime stamps (two workgroups, waves 0 / 1, 64 tiles, 8 slots);
 * host_out (may be NULL) receives the 2 x 2 x 64 x 8 stamps of the last traced launch */
int pa_attn_trace(int enable, unsigned long long* host_out);
int pa_attn_fwd(int dtype, const void* qkv, int64_t ldq, const void* rcat, void* out, int64_t ldo, float* lse,
                void* tables, int batch, int L, int heads, int Hp, int Wp, int head_dim, float scale, hipStream_t stream);

/* autograd of pa_attn_fwd (no reference source: torch autograd of the lines above).
 *   delta  : f32 [batch*heads, L] = rowsum(dO o O)                      (pa_attn_bwd_delta)
 *   dqkv   : T, same layout as qkv (dq | dk | dv)
 *   dG     : T [batch*L, heads*NRP] r-space bias gradient, consumed by pa_attn_bwd_relpos (NULL when relpos_part is given)
 *   relpos_part : NULL, or pa_attn_bwd_relpos_partials_bytes() (> 0 only where the 28-token-wide bf16 kernels run and `tables` is
 *            given) of device scratch: the dQ kernel then contracts d rel_pos itself -- one fp32 [NRP, hd] partial per workgroup --
 *            dG is not written, and pa_attn_bwd_relpos_reduce() sums the partials into drcat in a fixed order (deterministic)
 *   aux    : scratch of pa_attn_bwd_aux_bytes()
 *   tables : what pa_attn_fwd wrote (NULL: the backward recomputes the bias tables itself, generation-2 kernels); since ABI 5 the forward
 *            also writes the log-sum-exp fields of the tiles, the backward adds the Delta field.
 *            CONTRACT: the tiles must come from pa_attn_fwd of THIS library version (ABI >= 5) on the SAME qkv / rcat / scale -- the dKV kernel
 *            contracts the bf16 bias entries and the -lse / scale hi + lo fields it finds there and nothing checks their origin; tiles
 *            built or copied another way (the ABI-4 contract: a prep launch filled every field) give silently wrong dK / dV.  Callers
 *            that cannot guarantee it pass tables = NULL (generation 2 recomputes everything) or run pa_attn_bwd_prep, which rewrites the
 *            lse and Delta fields from `lse` / `out` / `dout` (the bias entries still have to be the forward's).
 *   out / ldo : the forward's output O (T [batch*L, heads*hd]) or NULL.  Given with delta = NULL where pa_attn_bwd_prep_ok(): the dQ kernel
 *            computes Delta = rowsum(dO o O) itself -- no pa_attn_bwd_delta / pa_attn_bwd_prep launch at all (ABI 5; the engine's route)
 *   rcatT  : T [hd, NRP] from pa_relpos_pack_t
 *   drcat  : f32 [NRP, hd] = [d rel_pos_h ; d rel_pos_w ; 0], overwritten */
int pa_relpos_pack_t(int dtype, const float* rel_pos_h, const float* rel_pos_w, void* rcatT, int Hp, int Wp, int head_dim,
                     hipStream_t stream);
int pa_attn_bwd_delta(int dtype, const void* out, int64_t ldo, const void* dout, int64_t lddo, float* delta, int batch,
                      int L, int heads, int head_dim, hipStream_t stream);
/* pa_attn_bwd_prep_ok() == 1 (28-token-wide bf16 kernels, `tables` given): Delta can live in the table tiles.  Either pa_attn_bwd is given
 * `out` and delta = NULL (the dQ kernel computes Delta: no extra launch), or -- the round-4 route, kept for A/B and tests -- pa_attn_bwd_prep
 * computes Delta and writes it with the log-sum-exp fields into the tiles in one pass and pa_attn_bwd is called with delta = out = NULL. */
int pa_attn_bwd_prep_ok(int dtype, int L, int Hp, int Wp, int head_dim);
int pa_attn_bwd_prep(int dtype, const void* out, int64_t ldo, const void* dout, int64_t lddo, const float* lse, void* tables,
                     int batch, int L, int heads, int Hp, int Wp, int head_dim, float scale, hipStream_t stream);
int64_t pa_attn_bwd_aux_bytes(int batch, int L, int heads, int Hp, int Wp);
int64_t pa_attn_bwd_relpos_partials_bytes(int dtype, int batch, int L, int heads, int Hp, int Wp, int head_dim);
int pa_attn_bwd(int dtype, const void* qkv, int64_t ldq, const void* rcat, const void* rcatT, const void* dout,
                int64_t lddo, const float* lse, const float* delta, void* dqkv, void* dG, void* relpos_part, void* aux,
                void* tables, const void* out, int64_t ldo, int batch, int L, int heads, int Hp, int Wp, int head_dim, float scale,
                hipStream_t stream);
int64_t pa_attn_bwd_relpos_workspace_bytes(int dtype, int batch, int L, int heads, int Hp, int Wp, int head_dim);
int pa_attn_bwd_relpos_reduce(const void* relpos_part, float* drcat, void* workspace, int batch, int L, int heads, int Hp, int Wp,
                              int head_dim, hipStream_t stream);
int pa_attn_bwd_relpos(int dtype, const void* dG, const void* qkv, int64_t ldq, float* drcat, void* workspace,
                       int batch, int L, int heads, int Hp, int Wp, int head_dim, hipStream_t stream);

/* ---- PatchEmbed (Conv2d k=P s=P as an im2col GEMM) + token assembly: util/vitdet_utils.py:182-186,
 *      models_painter.py:387-409 (mask token, segment tokens, abs pos), models_seggpt.py:415-420 (type tokens) ----
 * imgs/tgts: f32 NCHW [B,3,Hp*P,Wp*P]; w: T [D, ldw], ldw >= Kp = 3*P*P rounded up to 8, columns >= 3*P*P zero (pa_patch_weight_pack
 * makes it from the conv weight; for P % 8 == 0 it is the plain T copy with ldw = 3*P*P); any P >= 1 (P = 14: ViT-H/14);
 * pos: f32 [L, D] from pa_pos_fwd; mask: bool bytes [B or 1, L];
 * type_cls/type_ins/seg_type: SegGPT only (NULL otherwise), seg_type f32 [B];
 * tokens: f32 [2*B*L, D] = cat(x stream, y stream) on the batch axis (models_painter.py:409). */
int pa_patch_weight_pack(int dtype, const float* w /*f32 [D, 3*P*P]*/, void* out /*T [D, Kp]*/, int D, int P, hipStream_t stream);
int pa_patch_embed_fwd(int dtype, const float* imgs, const float* tgts, const void* w, int64_t ldw, const float* bias,
                       const float* mask_token, const float* seg_x, const float* seg_y, const float* pos,
                       const unsigned char* mask, int mask_batch_stride, const float* type_cls, const float* type_ins,
                       const float* seg_type, float* tokens, int batch, int Hp, int Wp, int P, int D, hipStream_t stream);
/* bf16 fast path (P % 8 == 0 and 3*P*P % 128 == 0, i.e. every reference factory): materialise the im2col operand once
 * (cols: bf16 [2*B*L, 3*P*P], also the X operand of the weight gradient: pa_linear_wgrad(dpe, cols)) and run the contraction on the
 * 256x256 LDS-DMA kernel.  pa_patch_cols_ok() says whether the shape qualifies; pa_patch_embed_fwd / _wgrad serve every other case. */
int pa_patch_cols_ok(int batch, int L, int P, int D);
int pa_patch_im2col(const float* imgs, const float* tgts, void* cols, int batch, int Hp, int Wp, int P, hipStream_t stream);
int pa_patch_embed_fwd_cols(const void* cols, const void* w /*bf16 [D, ldw]*/, int64_t ldw, const float* bias, const float* mask_token,
                            const float* seg_x, const float* seg_y, const float* pos, const unsigned char* mask, int mask_batch_stride,
                            const float* type_cls, const float* type_ins, const float* seg_type, float* tokens, int batch, int L, int K,
                            int D, hipStream_t stream);
int64_t pa_patch_embed_wgrad_workspace_bytes(int D, int P);
int pa_patch_embed_wgrad(int dtype, const void* dpe /*T [2BL, D]*/, const float* imgs, const float* tgts,
                         float* dw /*f32 [D, 3*P*P]*/, void* workspace, int batch, int Hp, int Wp, int P, int D,
                         hipStream_t stream);
/* get_abs_pos (util/vitdet_utils.py:128-157) as the constant bicubic operator M [L, S] (host-built), in row-sparse form
 * (painter_amd/hostmath.py sparse_rows: int32 column indices + f32 values, K entries per row, zero-padded): pos = M . pe with
 * (idx, val, K) of M; dpe = M^T . (gx + gy) with those of M^T.  pe/dpe point at pos_embed[0, skip_cls:, :] ([S, D]). */
int pa_pos_fwd(const int* idx /*[L, K]*/, const float* val /*[L, K]*/, int K, const float* pe /*[S, D]*/, float* pos /*[L, D]*/, int L, int D,
               hipStream_t stream);
int pa_pos_bwd(const int* idxT /*[S, KT]*/, const float* valT /*[S, KT]*/, int KT, const float* gx /*[L, D]*/, const float* gy /*[L, D]*/,
               float* dpe /*[S, D]*/, int S, int D, hipStream_t stream);
/* backward of the token assembly: dpe T [2BL, D]; sums f32 [3, L, D] = sum_b dx | sum_b dy | sum_b w*dy */
int pa_tokens_bwd(int dtype, const float* dx0, const unsigned char* mask, int mask_batch_stride, void* dpe, float* sums,
                  int batch, int L, int D, hipStream_t stream);

/* ---- x = (x[:B] + x[B:]) * 0.5 after block merge_idx (models_painter.py:414-415) and its backward ---- */
int pa_merge_fwd(const float* x, float* out, int64_t n_out, hipStream_t stream);
int pa_merge_bwd(int dtype, const float* dmerged, float* dx, void* dxT, const float* rowscale, int rows_per_sample,
                 int64_t rows_half, int D, hipStream_t stream);
int pa_scale_cast(int dtype, const float* in, void* out, const float* rowscale, int rows_per_sample, int64_t rows, int D,
                  hipStream_t stream);
int pa_cast_bf16(const float* in, void* out, int64_t n, hipStream_t stream);
/* diagnostics only (tools/gradsync_overlap.py): scratch = 0.5 * (scratch + src), `passes` times, on exactly `nblocks` persistent
 * workgroups -- a stand-in for the CU / HBM footprint of a ring all-reduce with `nblocks` channels.  Never on the product path. */
int pa_debug_rmw(const float* src, float* scratch, int64_t n, int passes, int nblocks, hipStream_t stream);
/* SegGPT cross-prompt feature ensemble + residual (models_seggpt.py:220-232): x1 = x0 + ens(a), groups of `group` samples */
int pa_ensemble_resid(const float* x0, const float* a, float* x1, int batch, int group, int L, int D, hipStream_t stream);

/* ---- decoder_pred: Conv3x3(64->64) -> LayerNorm2D -> GELU -> Conv1x1(64->3), one fused kernel
 *      (models_painter.py:328-333,:430; util/vitdet_utils.py:204-209) and its backward pieces ---- */
int pa_conv3x3_pack(int dtype, const float* w3 /*[64,64,3,3]*/, void* w3r /*T [64][9][64]*/, void* wf /*T [64][9][64]*/,
                    hipStream_t stream);
int pa_decoder_tail_fwd(int dtype, const void* x_nhwc, const void* w3r, const float* b3, const float* ln_gamma,
                        const float* ln_beta, const float* w1 /*[3,64]*/, const float* b1, void* y3 /*T NHWC or NULL*/,
                        float* pred /*f32 NCHW [B,3,Hi,Wi]*/, int batch, int Hi, int Wi, float eps, hipStream_t stream);
int64_t pa_decoder_tail_bwd_workspace_bytes(int batch, int Hi, int Wi);
/* grads: f32 [324] = dgamma[64] | dbeta[64] | dW1[3*64] | db1[3] | pad; dy3: T NHWC gradient of the conv3x3 output */
int pa_decoder_tail_bwd_pointwise(int dtype, const float* dpred, const void* y3, const float* ln_gamma,
                                  const float* ln_beta, const float* w1, void* dy3, float* grads, void* workspace,
                                  int batch, int Hi, int Wi, float eps, hipStream_t stream);
/* dE: T [B*Hp*Wp, P*P*64] = gradient w.r.t. decoder_embed's output (pixel shuffle inverted in the epilogue) */
int pa_conv3x3_dgrad_unshuffle(int dtype, const void* dy3, const void* wf, void* dE, int batch, int Hp, int Wp, int P,
                               hipStream_t stream);
int64_t pa_conv3x3_wgrad_workspace_bytes(int batch, int Hi, int Wi);
int pa_conv3x3_wgrad(int dtype, const void* dy3, const void* x_nhwc, float* dw /*f32 [64,64,3,3]*/, void* workspace,
                     int batch, int Hi, int Wi, hipStream_t stream);

/* ---- forward_loss (models_painter.py:433-462; SegGPT models_seggpt.py:448-469) ----
 * out: f32 [2] = {loss, denominator}.  ignore_rule = 1 (Painter): samples whose unmasked de-normalised target sums
 * below 300 get valid := 0 IN PLACE (models_painter.py:444-448).  kind: 0 smoothl1(beta) 1 l1 2 l2 3 l1l2. */
int64_t pa_loss_workspace_bytes(int batch, int Hi, int Wi);
int pa_loss_fwd(const float* pred, const float* tgts, float* valid, const unsigned char* mask, int mask_batch_stride,
                float* out, void* workspace, int batch, int Hi, int Wi, int P, int ignore_rule, float eps_den, int kind,
                float beta, hipStream_t stream);
int pa_loss_bwd(const float* pred, const float* tgts, const float* valid, const unsigned char* mask,
                int mask_batch_stride, const float* dloss, const float* loss_out, float* dpred, int batch, int Hi, int Wi,
                int P, int kind, float beta, hipStream_t stream);
/* patchify (models_painter.py:355-368): f32 NCHW -> [B, L, P*P*3] */
int pa_patchify(const float* img, float* out, int batch, int Hp, int Wp, int P, hipStream_t stream);

/* ---- Optimizer step: GradScaler.unscale_ + clip_grad_norm_ + AdamW over the layer-decay groups in two passes.
 * Replaces Painter/util/misc.py:256-268 (NativeScalerWithGradNormCount.__call__) acting on the torch.optim.AdamW of
 * Painter/main_train.py:344-348 (parameter groups: util/lr_decay.py:15-61; schedule: util/lr_sched.py:9-21).
 * table: DEVICE array of ntensors records, first_chunk = running sum of ceil(n / pa_opt_chunk_elems()); nchunks = their total.
 * g == NULL marks a parameter without gradient (skipped).  All tensors fp32, contiguous. */
typedef struct PaOptTensor {
    void* p; const void* g; void* m; void* v;    /* parameter, gradient, exp_avg, exp_avg_sq */
    void* w16;                                   /* optional bf16 copy of p refreshed by pa_adamw_step, or NULL */
    int64_t n;                                   /* elements */
    int32_t group;                               /* index into PaOptGroups */
    int32_t first_chunk;
} PaOptTensor;
typedef struct PaOptGroups {                     /* HOST struct, passed by value to the kernel: at most 64 parameter groups */
    float lr[64], wd[64];                        /* learning rate (already times lr_scale), weight decay */
    int32_t active[64];                          /* 1 = the group has gradients in this step (its step count advances) */
} PaOptGroups;
int pa_opt_chunk_elems(void);
int64_t pa_grad_sumsq_workspace_bytes(int nchunks);
/* out2 (device, 2 floats): [0] = sum over all gradients of g^2 (as stored, i.e. still multiplied by the loss scale),
 * [1] = 0 if every gradient is finite */
int pa_grad_sumsq(const PaOptTensor* table, int ntensors, int nchunks, float* out2, void* workspace, hipStream_t stream);
/* steps: DEVICE float[64], the per-group AdamW step counts (advanced here unless the step is skipped);
 * norm_info = out2 of pa_grad_sumsq or NULL (no clipping / no finiteness gate); grad_scale = device scalar the gradients are
 * divided by, or NULL; found_inf = device scalar, nonzero skips the step (torch GradScaler protocol), or NULL;
 * max_norm <= 0 disables clipping.  groups is a HOST pointer. */
int pa_adamw_step(const PaOptTensor* table, int ntensors, int nchunks, const PaOptGroups* groups, float beta1, float beta2,
                  float eps, float* steps, const float* norm_info, const float* grad_scale, const float* found_inf,
                  float max_norm, hipStream_t stream);

/* ---- SegGPT pre-/post-processing on the device.  Replaces the PIL / numpy / CPU-torch work of
 * SegGPT/SegGPT_inference/seggpt_engine.py: inference_image :56-103, inference_video :106-181, run_one_image :26-53.
 * Images are uint8 [H][W][C] (C <= 4), densely packed, in device memory; everything is bit-exact with the reference's host path:
 * integer arithmetic for the resize passes, float64 with one rounding per operation for normalise / de-normalise / blend. ---- */
/* One separable pass of Pillow's ImagingResample for 8-bit pixels (`Image.resize`, default BICUBIC; :62, :66, :117, :136):
 * out = clip8((2^21 + sum_t src[first + t] * coeffs[o][t]) >> 22) along the width (vertical = 0: dst is [src_h][dst_w]) or the
 * height (vertical = 1: dst is [dst_h][src_w]).  bounds: int32 [n_out][2] = (first, taps); coeffs: int32 [n_out][ksize]. */
int pa_resample_u8(const void* src, int src_h, int src_w, void* dst, int dst_h, int dst_w, int channels, const void* bounds,
                   const void* coeffs, int ksize, int vertical, hipStream_t stream);
/* `Image.resize(size, Image.NEAREST)` (:70, :121) and any other index-table gather: dst[y][x] = src[ytab[y]][xtab[x]], 0 where a
 * table entry is negative.  ytab: int32 [dst_h], xtab: int32 [dst_w]. */
int pa_gather_u8(const void* src, int src_h, int src_w, void* dst, int dst_h, int dst_w, int channels, const void* ytab,
                 const void* xtab, hipStream_t stream);
/* :73-92 + :28-34: imgs[n] = [prompt_n ; query], tgts[n] = [target_n ; target_n] along H, (v / div - mean) / std in float64, written
 * as float32 NCHW [n_prompts][3][2*res_h][res_w].  prompts / targets: uint8 [n_prompts][res_h][res_w][3]; query: uint8
 * [res_h][res_w][3]; target_div: DEVICE double [n_prompts] (255 for image-scale targets, 1 for the cached {0,1} masks, :166-171). */
int pa_seggpt_stitch(const void* prompts, const void* targets, const void* target_div, const void* query, float* imgs,
                     float* tgts, int n_prompts, int res_h, int res_w, hipStream_t stream);
/* :49-53: pred = the model's float32 tokens of sample 0, [2*res_h/patch * res_w/patch][patch*patch*3]; out = float64
 * [res_h][res_w][3] = clip((lower half of unpatchify(pred) * std + mean) * 255, 0, 255). */
int pa_seggpt_decode(const float* pred, void* out_f64, int res_h, int res_w, int patch, hipStream_t stream);
/* :166-171: out uint8 [res_h][res_w][3] = (mean over channels of the decoded picture > 128), the next video prompt target. */
int pa_seggpt_mask(const float* pred, void* out_u8, int res_h, int res_w, int patch, hipStream_t stream);
/* :95-102, :173-179 fused: decode, nearest-resize to out_h x out_w through ytab / xtab (int32 source rows / columns of the
 * res_h x res_w picture, F.interpolate(mode='nearest') rule), out = uint8(image * (0.6 * picture / 255 + 0.4)) (truncation).
 * image, out: uint8 [out_h][out_w][3]. */
int pa_seggpt_blend(const float* pred, const void* image, void* out, int out_h, int out_w, const void* ytab, const void* xtab,
                    int res_h, int res_w, int patch, hipStream_t stream);

/* ---- Painter training input pipeline, pixel work on the device.  Replaces, per sample, the PIL / CPU-torch work of
 * Painter/data/pairdataset.py:106-190 under the transform stack of Painter/main_train.py:232-251 (Painter/data/pair_transforms.py).
 * Random parameters (crop boxes, jitter order and factors, flip flags) are inputs: drawing them stays on the host. ---- */
/* RandomResizedCrop on decoded uint8 pictures (pair_transforms.py:152-163 -> PIL crop + resize): pa_resample_u8 / pa_gather_u8 on a
 * box of a larger picture -- src points at the box's first pixel, rows are src_row_bytes apart. */
int pa_resample_u8_box(const void* src, int64_t src_row_bytes, int src_h, int src_w, void* dst, int dst_h, int dst_w, int channels,
                       const void* bounds, const void* coeffs, int ksize, int vertical, hipStream_t stream);
int pa_gather_u8_box(const void* src, int64_t src_row_bytes, int src_h, int src_w, void* dst, int dst_h, int dst_w, int channels,
                     const void* ytab, const void* xtab, hipStream_t stream);
/* ColorJitter (pair_transforms.py:236-247 -> PIL ImageEnhance.Brightness / Contrast / Color and the HSV hue shift), in place on uint8
 * [batch][h][w][3].  ops: DEVICE int32 [batch][4], the op of each of the four slots in applied order (0 brightness, 1 contrast,
 * 2 saturation, 3 hue, negative = none); factors: DEVICE float [batch][4] (hue slot: the uint8 added to the H plane, as a float);
 * ops_host: HOST copy of ops, or NULL (then every slot and its mean pass is launched).  Bit-exact with Pillow. */
int64_t pa_color_jitter_workspace_bytes(int batch);
int pa_color_jitter(void* images, const void* ops, const void* factors, const void* ops_host, void* workspace, int batch, int h, int w,
                    hipStream_t stream);
/* RandomHorizontalFlip + ToTensor + Normalize (pair_transforms.py:199-203, :72, :101) and the two-pair stitch (pairdataset.py:100-104):
 * uint8 [batch][h][w][3] -> rows [row0, row0 + h) of float32 [batch][3][canvas_h][w]; flip: DEVICE int32 [batch]. */
int pa_to_tensor_normalize(const void* images, const void* flip, float* canvas, int batch, int h, int w, int canvas_h, int row0,
                           hipStream_t stream);
/* Second RandomResizedCrop on the float canvases (main_train.py:248-250): dst[b] = interpolate(src[b][:, top:top+bh, left:left+bw],
 * size = (h, w)), bicubic (A = -0.75, align_corners = False, no antialias) or nearest.  boxes: DEVICE int32 [batch][4] =
 * (top, left, bh, bw).  src != dst. */
int pa_resized_crop_f32(const float* src, float* dst, const void* boxes, int batch, int channels, int h, int w, int nearest,
                        hipStream_t stream);
/* The same with a per-sample mode: modes = DEVICE int32 [batch], 0 bicubic, 1 nearest, 2 the sample keeps its canvas (plain copy): the
 * samples of a step that take no second crop, or different interpolations (pairdataset.py:113-124), go through one launch. */
int pa_resized_crop_f32_modes(const float* src, float* dst, const void* boxes, const void* modes, int batch, int channels, int h, int w,
                              hipStream_t stream);
/* RandomResizedCrop (pair_transforms.py:152-163 -> PIL crop + resize) for every decoded picture of a step in two launches.  jobs: DEVICE
 * array of n_jobs descriptors; every pointer in a job is a device address.  Bicubic jobs: xbounds / ybounds = int32 [out][2] (first
 * tap, taps), xcoeffs / ycoeffs = int32 [out][ksize] -- Pillow's fixed-point tables (painter_amd.hostmath / RS.bicubic_tables), a pass
 * whose size does not change is skipped as in Pillow and its tables may be NULL.  Nearest jobs: xbounds = int32 [out_w] source column,
 * ybounds = int32 [out_h] source row (PIL nearest tables), unused when the size does not change.  mid: DEVICE scratch of
 * (sum of the bicubic jobs' h) * out_w * 3 bytes; a job's rows start at row mid_row0.  max_h / max_w: the largest box.  Bytes are
 * identical to pa_resample_u8_box / pa_gather_u8_box applied picture by picture. */
typedef struct pa_crop_job {
    const void* src;            /* first pixel of the crop box, uint8 RGB, rows src_row_bytes apart */
    void* dst;                  /* uint8 [out_h][out_w][3] */
    const void* xbounds;
    const void* xcoeffs;
    const void* ybounds;
    const void* ycoeffs;
    int64_t src_row_bytes;
    int32_t h, w;               /* box size */
    int32_t xksize, yksize;
    int32_t nearest;            /* 0 bicubic, 1 PIL nearest */
    int32_t mid_row0;
} pa_crop_job;
int pa_resized_crop_u8_batch(const pa_crop_job* jobs, void* mid, int n_jobs, int max_h, int max_w, int out_h, int out_w, hipStream_t stream);
/* `valid` rules (pairdataset.py:152-180) on float32 [batch][3][plane] targets.  modes: DEVICE int32 [batch] (0 ones; 1 target < thres
 * -> 0; 2 target > thres -> 10 and all 0 if fewer than 300 foreground elements; 3 all 0 if fewer than 300 foreground elements);
 * thres: DEVICE float [batch][3]. */
int64_t pa_pair_valid_workspace_bytes(int batch);
int pa_pair_valid(const float* tgts, float* valid, const void* modes, const void* thres, void* workspace, int batch, int plane,
                  hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PAINTER_HIP_H */
