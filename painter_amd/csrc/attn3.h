// entry points of the third-generation bf16 attention kernels (attn3.hip: key rows of 28 tokens, i.e. the 896x448 / patch 16 grid),
// used by pa_attn_fwd / pa_attn_bwd
#pragma once
#include "common.h"
bool attn3_ok(int L, int Hp, int Wp);
// per-(sample, head, 32-query tile) table tiles written by the forward and read by the backward; 0 when attn3_ok() is false
int64_t attn3_table_bytes(int Bn, int L, int H, int Hp, int Wp);
int attn3_fwd(const bf16* qkv, int64_t ldq, const bf16* rcat, bf16* out, int64_t ldo, float* lse, void* tables, int Bn, int L, int H,
              int Hp, int Wp, float scale, hipStream_t st);
// part != NULL (attn3_relpos_partials_bytes() of fp32 scratch): the dQ kernel contracts the rel-pos table gradient itself and writes one
// partial per workgroup there instead of dG; attn3_relpos_reduce() sums the partials into drcat [NRP][64]
// out / ldo (may be NULL / 0, only read when delta == NULL): the forward's output -- the dQ kernel then computes Delta itself (no prep launch)
int attn3_bwd(const bf16* qkv, int64_t ldq, const bf16* rcatT, const bf16* dout, int64_t lddo, const float* lse, const float* delta,
              void* tables, bf16* dqkv, bf16* dG, float* part, int Bn, int L, int H, int Hp, int Wp, float scale, const bf16* out, int64_t ldo,
              hipStream_t st);
// Delta = rowsum(dO o O) computed and written (with -lse / scale hi + lo) straight into the table tiles: replaces pa_attn_bwd_delta + the prep
// kernel of attn3_bwd, which then takes delta = NULL
int attn3_bwd_prep(const bf16* out, int64_t ldo, const bf16* dout, int64_t lddo, const float* lse, void* tables, int Bn, int L, int H, int Hp,
                   float scale, hipStream_t st);
int64_t attn3_relpos_partials_bytes(int Bn, int L, int H, int Hp, int Wp);       // 0: not fused for this grid (or PA_ATTN3_FUSE_RELPOS=0)
int attn3_relpos_reduce(const float* part, float* drcat, float* tmp, int Bn, int L, int H, int Hp, int Wp, hipStream_t st);
