// Pieces of the generation-3 attention kernels (attn3.hip, 4-wave workgroups; the retired paired 8-wave and software-pipelined builds
// under tools/experiments/ used them too).  See attn3.hip for the scheme (rel-pos bias as a one-hot contraction).
#pragma once
#include "attn_tile.h"
#include <type_traits>

namespace a3 {
using namespace atile;

constexpr int WP = 28, PH = 7, RPP = 8;       // key-row width; 32-key tile phases per period; key rows per period (7 * 32 = 8 * 28)
constexpr int EIMG = 2048;                    // one one-hot image: [32 keys][32 slots] bf16
constexpr float THR = 6.0f;

__host__ __device__ inline int ttile_bytes(int Hp) { return (2048 + (Hp + 2) * 64 + 128 + 255) & ~255; }
// physical slot of kw inside a 32-slot T row / E row: k-step 0 = kw 0..15; k-step 1: half-wave g holds kw 16+6g .. 21+6g in t = 0..5 and
// the window slots 2g, 2g+1 in t = 6, 7
DEVI int kw_phys(int kw) { return kw < 22 ? kw : kw + 2; }

// one-hot images of the 7 tile phases.  Row i = key 32 p + i of a period: kw = (4 p + i) % 28, key row (relative) p + ((4 p + i) >= 28).
// 16-byte chunk c = 2 s + g of a row is stored at c ^ ((i >> 2) & 3): conflict-free for the row reads (ds_read_b128) and for the
// transposing reads (which always see 4 consecutive rows of one 4-row group).
DEVI void build_eimg(unsigned char* eimg, int tid, int nthreads = NT) {
    // one thread per image row: 64 zero bytes, then the two ones (same-thread LDS writes are ordered)
    for (int row = tid; row < PH * 32; row += nthreads) {
        const int ph = row >> 5, i = row & 31;
        const int a = 4 * ph + i, kw = a >= WP ? a - WP : a, khr = ph + (a >= WP ? 1 : 0);        // a < 2 * WP
        const int sw = (i >> 2) & 3;
        unsigned char* r = eimg + ph * EIMG + i * 64;
#pragma unroll
        for (int c = 0; c < 4; ++c) *reinterpret_cast<uint4*>(r + c * 16) = zero4();
        const int e0 = kw_phys(kw);                                         // element = 8 * chunk + t
        const int e1 = 16 + 8 * ((khr & 3) >> 1) + 6 + (khr & 1);           // window slot khr & 3: half-wave (slot >> 1), t = 6 + (slot & 1)
        *reinterpret_cast<uint16_t*>(r + (((e0 >> 3) ^ sw) << 4) + (e0 & 7) * 2) = 0x3F80u;
        *reinterpret_cast<uint16_t*>(r + (((e1 >> 3) ^ sw) << 4) + (e1 & 7) * 2) = 0x3F80u;
    }
}
struct EAddr {
    int row[2];    // [32 rows][64 B] image, chunk-swizzled as above: 16-byte row fragment of k-step s (lane = row)
    int tr[2];     // transposed fragment (rows = slots, contraction over the image's rows): lo / hi; k-step 1 = + 1024
    DEVI void init(int lane) {
        const int i = lane & 31, g = lane >> 5;
#pragma unroll
        for (int s = 0; s < 2; ++s) row[s] = i * 64 + (((2 * s + g) ^ ((i >> 2) & 3)) << 4);
        const int ii = lane & 15, half = (lane >> 4) & 1;
#pragma unroll
        for (int hi = 0; hi < 2; ++hi) {
            const int r = 4 * g + (ii >> 2) + 8 * hi;
            tr[hi] = r * 64 + (((2 * half + ((ii & 3) >> 1)) ^ ((r >> 2) & 3)) << 4) + (ii & 1) * 8;
        }
    }
};
DEVI bf16x8 efrag(const unsigned char* img, const EAddr& e, int s) {
    return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(img + e.row[s]));
}
DEVI bf16x8 etrfrag(const unsigned char* img, const EAddr& e, int s) {
    const u32x2 l = ldtr(img + e.tr[0] + s * 1024), h = ldtr(img + e.tr[1] + s * 1024);
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(l, h, 0, 1, 2, 3));
}
DEVI bf16x8 as_frag(const uint4& v) { return __builtin_bit_cast(bf16x8, v); }
DEVI f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    return z;
}
// window slot update: 16 bits of `w` <- the kh-table entry at p (every lane of the wave, see the file header)
template <int HALF> DEVI void win_set(uint32_t& w, const unsigned char* p) {
    const uint32_t v = *reinterpret_cast<const uint16_t*>(p);
    w = HALF ? ((w & 0xffffu) | (v << 16)) : ((w & 0xffff0000u) | v);
}
// eacc register (half-wave 1) that holds window slot `slot`: rows 22, 23, 30, 31 of the D tile
DEVI constexpr int win_reg(int slot) { return slot == 0 ? 10 : slot == 1 ? 11 : slot == 2 ? 14 : 15; }

// T of this lane's query row from G^T = Rcat . Q^T: kw part -> twimg[q][32 slots] (bf16, window slots stay zero),
// kh part -> thT[kh][q] (bf16).  Both scaled by 1 / scale, so that logits = scale * log2e * (q.k + T.E).
DEVI void build_tables3(unsigned char* twimg, unsigned char* thT, const bf16* rcat, int NRP, const bf16x8 (&qf)[4], int qh, int qw, int Hp,
                        float inv_scale, int lane, unsigned char* trash) {
    // The scatter of G^T's D tile into the two tables is straight-line code: per accumulator register one convert, one range test, one
    // select between the entry's address and this lane's `trash` halfword (2 bytes per lane of LDS nobody reads), one ds_write_b16.
    // With a branch per element the table build was 30 us of the 145 us forward launch at the ViT-L grid (ablation, tools/attn_ablate.py).
    const int g = lane >> 5, ql = lane & 31;
    *reinterpret_cast<uint4*>(twimg + lane * 32) = zero4();
    *reinterpret_cast<uint4*>(twimg + lane * 32 + 16) = zero4();
    const int nkh = 2 * Hp - 1;
    unsigned char* const th0 = thT + ql * 2;
    unsigned char* const tw0 = twimg + ql * 64;
    for (int rbk = 0; rbk < NRP / 32; ++rbk) {
        f32x16 acc = zero16();
        const bf16* rp = rcat + (size_t)(rbk * 32 + ql) * ATT_HD;
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = mfma(gfrag(rp, s, g), qf[s], acc);
        const int r0 = rbk * 32 + 4 * g;                            // r of register 0; register reg adds (reg & 3) + 8 * (reg >> 2)
        if (rbk * 32 < nkh) {                                       // block holds rel_pos_h rows (wave-uniform)
            const int kh0 = qh + Hp - 1 - r0;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int off = (reg & 3) + 8 * (reg >> 2);
                const int kh = kh0 - off;
                const bool ok = (unsigned)kh < (unsigned)Hp;          // implies r < 2 Hp - 1 (r = qh + Hp - 1 - kh, qh < Hp); rows of the w part give kh < 0
                *reinterpret_cast<bf16*>(ok ? th0 + kh * 64 : trash) = (bf16)(acc[reg] * inv_scale);
            }
        }
        if (rbk * 32 + 32 > nkh) {                                  // block holds rel_pos_w rows
            const int kw0 = qw + WP - 1 - (r0 - nkh);
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int off = (reg & 3) + 8 * (reg >> 2);
                const int kw = kw0 - off;                           // = qw + WP - 1 - rr
                const bool ok = (unsigned)kw < (unsigned)WP;          // implies 0 <= rr < 2 WP - 1; h-part rows give kw >= WP, padding rows kw < 0
                *reinterpret_cast<bf16*>(ok ? tw0 + kw_phys(kw) * 2 : trash) = (bf16)(acc[reg] * inv_scale);
            }
        }
    }
}


// -lse / scale as a bf16 hi + lo pair and -Delta into the table tiles (one thread per (sample, head, query))
__global__ void prep_kernel(const float* __restrict__ lse, const float* __restrict__ delta, unsigned char* __restrict__ tables, int L, int Hp,
                            float inv_scale, int total);

}   // namespace a3
