// entry points of the second-generation bf16 attention kernels (attn2.hip), used by pa_attn_fwd / pa_attn_bwd
#pragma once
#include "common.h"
bool attn2_ok(int L, int Hp, int Wp, int hd = 64);       // hd: 64, or 80 (ViT-H/14: also key rows of 32 tokens)
int attn2_fwd(const bf16* qkv, int64_t ldq, const bf16* rcat, bf16* out, int64_t ldo, float* lse, int Bn, int L, int H, int Hp, int Wp,
              int hd, float scale, hipStream_t st);
int64_t attn2_aux_bytes(int Bn, int L, int H, int Hp, int Wp);
int attn2_bwd(const bf16* qkv, int64_t ldq, const bf16* rcat, const bf16* rcatT, const bf16* dout, int64_t lddo, const float* lse,
              const float* delta, bf16* dqkv, bf16* dG, void* aux, int Bn, int L, int H, int Hp, int Wp, int hd, float scale, hipStream_t st);
