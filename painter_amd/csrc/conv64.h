// bf16 3x3 convolution kernels for the decoder head (64 -> 64 channels, NHWC, pad 1), written for gfx950:
//   conv3x3_tile_kernel   out[px][co] = sum_{tap,ci} W[co][tap][ci] . X[px + off(tap)][ci]   (forward tail and data gradient)
//   conv3x3_wgrad_kernel  dW[co][ci][tap] = sum_px dY[px][co] . X[px + off(tap)][ci]
// Replaces the im2col gather operands of the generic engine for Painter/models_painter.py:328-333,:430:
// the gather re-read every input pixel 9 times from L2 in 16-byte pieces; here a workgroup stages one halo tile of the image in
// LDS once ([pixel][64 ch] = 128-byte rows, 16-byte chunk index XOR-swizzled with a bijection of pixel-column bits 1..3 so that
// both ds_read_b128 pixel fragments and the transposing ds_read_b64_tr_b16 are bank-conflict free at every tap shift) and
// every tap is an immediate-offset LDS read of it.
//   forward / dgrad: workgroup = 4 waves = 4 image rows x 64 pixels, wave = one row (2 segments of 32 pixels) x 64 output
//     channels; the 9 taps' weights stream through an 8 KB double buffer; 144 MFMA 32x32x16 per wave; 2 workgroups per CU.
//     MFMA orientation A = weights (i = output channel), B = pixels (j): a lane owns one pixel and 32 of its channels, which is
//     what the fused LayerNorm2D / GELU / 1x1-conv epilogue wants.
//   wgrad: persistent workgroups (2 per CU) that walk down 64-pixel column strips in 2-row steps (ring of four halo rows in LDS: every
//     input byte is fetched once); wave = (output-channel half, input-channel half), 9 tap accumulators; the contraction runs over
//     pixels, so both operands are read with the transposing LDS read; the next step's global loads are in flight under the current
//     step's 72 MFMAs; per-workgroup fp32 slabs, reduced in fixed order.
#pragma once
#include "common.h"
#include <type_traits>

namespace c64 {

constexpr int TW = 64;                  // tile width in pixels (both kernels)
constexpr int HS = 80;                  // LDS pixels per halo row (>= TW + 2, multiple of 16 so the swizzle depends on the column only)
constexpr int FTH = 4;                  // forward tile height
constexpr int F_XB = (FTH + 2) * HS * 128, F_WB = 8192, F_LDS = F_XB + 2 * F_WB;
constexpr int WTH = 2;                  // wgrad tile height
constexpr int W_XB = (WTH + 2) * HS * 128, W_YB = WTH * TW * 128, W_LDS = W_XB + W_YB;
constexpr int W_SLAB = 64 * 64 * 9;     // floats per workgroup slab

template <int V> using IC = std::integral_constant<int, V>;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

DEVI int vsw(int r) { return (((r >> 1) & 1) << 2) | (((r >> 2) & 1) << 1) | ((r >> 3) & 1); }
DEVI int xbyte(int pix, int chunk) { return pix * 128 + ((chunk ^ vsw(pix)) << 4); }
DEVI f32x16 mfma(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
DEVI bf16x8 ld_frag(const unsigned char* p) { return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(p)); }
// transposed fragment: 8 contraction values (pixels 4g + (t & 3) + 8 (t >> 2) of the 16-pixel step) for row (lane & 31)
DEVI bf16x8 tr_frag(const unsigned char* lo, const unsigned char* hi) {
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)lo);
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)hi);
    const u32x2 l = __builtin_bit_cast(u32x2, a), h = __builtin_bit_cast(u32x2, b);
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(l, h, 0, 1, 2, 3));
}

// Epilogue stores of the tile kernel (round 6).  A lane owns one pixel and 4-channel runs of it (MFMA D layout), so storing from registers is
// 16 eight-byte stores per 32-pixel block, each touching 32 different 128-byte pixel rows -- store-issue-bound (the same finding as the GEMM
// and attention epilogues).  Instead the wave passes a [32 px][64 ch] bf16 block through 4 KB of LDS (16-byte chunk index XOR (px & 7): the
// 8-byte writes of a half-wave spread over all banks, the 16-byte reads of 8 lanes cover one row conflict-free) and stores whole pixel rows:
// 4 sixteen-byte stores per block, 8 lanes per 128-byte row.
DEVI void px_stage4(unsigned char* stg, int px, int c, uint2 v) {          // channels c .. c + 3 (c % 4 == 0) of pixel px (0..31)
    *reinterpret_cast<uint2*>(stg + px * 128 + (((c >> 3) ^ (px & 7)) << 4) + (c & 7) * 2) = v;
}
// row(px) -> destination of that pixel's 64 channels; valid(px) -> store it?
template <class RowFn> DEVI void px_write_rows(const unsigned char* stg, int lane, RowFn&& row) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int px = (lane >> 3) + 8 * i, ch = lane & 7;
        bf16* dst = row(px);
        if (dst != nullptr) *reinterpret_cast<uint4*>(dst + ch * 8) = *reinterpret_cast<const uint4*>(stg + px * 128 + ((ch ^ (px & 7)) << 4));
    }
}

// ------------------------------------------------------------------------------------------------ forward / dgrad
// x: NHWC [B, Hi, Wi, 64] bf16; w: [64 out][9 taps][64 in] bf16; Hi % 4 == 0, Wi % 64 == 0.
// Epi(acc[2][2], 0, first pixel (linear index) of the wave's 64-pixel run, lane, 0): acc[bi][bj][r] = out channel bi*32 + acc_row(r),
// pixel bj*32 + (lane & 31).
template <class Epi>
__global__ __launch_bounds__(256, 2) void conv3x3_tile_kernel(const bf16* __restrict__ x, const bf16* __restrict__ w, Epi epi, int Hi, int Wi) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 5, lr = lane & 31;
    const int ntx = Wi / TW, nty = Hi / FTH;
    int t = blockIdx.x;
    const int bx = t % ntx;
    t /= ntx;
    const int by = t % nty, b = t / nty;
    const int x0 = bx * TW, y0 = by * FTH;
    const bf16* img = x + (size_t)b * Hi * Wi * 64;

    constexpr int HC = TW + 2, NCH = (FTH + 2) * HC * 8, NLD = (NCH + 255) / 256;
    uint4 xr[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int idx = tid + 256 * i;
        xr[i] = zero4();
        if (idx < NCH) {
            const int pix = idx >> 3, ch = idx & 7, r = pix / HC, c = pix - r * HC;
            const int gy = y0 - 1 + r, gx = x0 - 1 + c;
            if (gy >= 0 && gy < Hi && gx >= 0 && gx < Wi) xr[i] = *reinterpret_cast<const uint4*>(img + ((size_t)gy * Wi + gx) * 64 + ch * 8);
        }
    }
    // The taps' weights go global -> registers -> LDS double buffer.  Round 6: the registers are loaded TWO taps ahead (two register pairs,
    // alternating): with one tap of cover (16 MFMAs = 512 cycles) every tap's LDS store sat waiting for an L2 round trip in front of the
    // tap's barrier -- nine exposed round trips per tile, matrix pipe busy 0.20.
    uint4 wa0, wa1, wb0, wb1;                        // (named registers: an indexed pair went to scratch)
    const int wco0 = tid >> 3, wch = tid & 7;       // weight chunk of this thread: rows wco0 and wco0 + 32
    auto load_w = [&](auto tap_c) {
        constexpr int tap = decltype(tap_c)::value;
        const uint4 v0 = *reinterpret_cast<const uint4*>(w + ((size_t)wco0 * 9 + tap) * 64 + wch * 8);
        const uint4 v1 = *reinterpret_cast<const uint4*>(w + ((size_t)(wco0 + 32) * 9 + tap) * 64 + wch * 8);
        if constexpr (tap & 1) { wb0 = v0; wb1 = v1; } else { wa0 = v0; wa1 = v1; }
    };
    auto store_w = [&](auto tap_c) {
        constexpr int tap = decltype(tap_c)::value;
        unsigned char* d = smem + F_XB + (tap & 1) * F_WB + wco0 * 128 + ((wch ^ vsw(wco0)) << 4);
        *reinterpret_cast<uint4*>(d) = (tap & 1) ? wb0 : wa0;
        *reinterpret_cast<uint4*>(d + 4096) = (tap & 1) ? wb1 : wa1;
    };
    load_w(IC<0>{});
    load_w(IC<1>{});
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int idx = tid + 256 * i;
        if (idx < NCH) {
            const int pix = idx >> 3, ch = idx & 7, r = pix / HC, c = pix - r * HC;
            *reinterpret_cast<uint4*>(smem + xbyte(r * HS + c, ch)) = xr[i];
        }
    }
    store_w(IC<0>{});
    __syncthreads();

    // fragment addresses: weights row (lane & 31) [+32 rows = +4096]; pixels: halo row `wave` (+ (dy+1) rows by immediate),
    // halo column dxi + (lane & 31) (+32 by immediate), chunk 2 s + g
    int wa[4], xa[3][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        wa[s] = F_XB + lr * 128 + (((2 * s + g) ^ vsw(lr)) << 4);
#pragma unroll
        for (int dxi = 0; dxi < 3; ++dxi) xa[dxi][s] = xbyte(wave * HS + dxi + lr, 2 * s + g);
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;

    auto do_tap = [&](auto tap_c) {
        constexpr int tap = decltype(tap_c)::value, dyi = tap / 3, dxi = tap % 3;
        const unsigned char* wimg = smem + (tap & 1) * F_WB;
        const unsigned char* ximg = smem + dyi * HS * 128;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bf16x8 a0 = ld_frag(wimg + wa[s]), a1 = ld_frag(wimg + wa[s] + 4096);
            const bf16x8 b0 = ld_frag(ximg + xa[dxi][s]), b1 = ld_frag(ximg + xa[dxi][s] + 32 * 128);
            acc[0][0] = mfma(a0, b0, acc[0][0]);
            acc[0][1] = mfma(a0, b1, acc[0][1]);
            acc[1][0] = mfma(a1, b0, acc[1][0]);
            acc[1][1] = mfma(a1, b1, acc[1][1]);
        }
        if constexpr (tap + 1 < 9) store_w(IC<tap + 1>{});    // (loaded during tap - 1; its register pair is free again behind this store)
        if constexpr (tap + 2 < 9) load_w(IC<tap + 2>{});
        __syncthreads();
    };
    do_tap(IC<0>{}); do_tap(IC<1>{}); do_tap(IC<2>{}); do_tap(IC<3>{}); do_tap(IC<4>{});
    do_tap(IC<5>{}); do_tap(IC<6>{}); do_tap(IC<7>{}); do_tap(IC<8>{});
    // (every tap ends with a barrier: the halo image and the weight buffers are dead -- 8 KB of staging per wave for the epilogue's stores)
    epi(acc, 0, (int)(((size_t)b * Hi + y0 + wave) * Wi + x0), lane, 0, smem + wave * 8192);
}

template <class Epi> static int launch_tile(const bf16* x, const bf16* w, Epi epi, int Bn, int Hi, int Wi, hipStream_t st) {
    auto kern = conv3x3_tile_kernel<Epi>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, F_LDS);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    PA_LAUNCH(kern, dim3(Bn * (Hi / FTH) * (Wi / TW)), dim3(256), F_LDS, st, x, w, epi, Hi, Wi);
    return (int)hipGetLastError();
}
static inline bool ok(int Bn, int Hi, int Wi) {
    return Hi % FTH == 0 && Hi % WTH == 0 && Wi % TW == 0 && (size_t)Bn * Hi * Wi < (1ull << 31);
}

// ------------------------------------------------------------------------------------------------ weight gradient
// Round 5 rewrite.  The round-1 kernel loaded a tile, stored it to LDS and only then multiplied (load -> barrier -> store -> barrier -> 72
// MFMAs, nothing in flight under the MFMAs: matrix pipe busy 0.12, waves waiting 0.83 of their cycles, 784 us for 237 GFLOP) and fetched
// every input row twice (a 2-row tile has a 4-row halo; vertically adjacent tiles ran on different workgroups): 1.23 GB per launch for
// 0.82 GB of operands.  Now:
//   * a workgroup walks DOWN the image: its tiles are consecutive 2-row steps of one 64-pixel column strip (linear tile index =
//     strip-major), the four halo rows live in a ring of four LDS row slots, and a step only fetches the two new rows (a fresh four-row
//     load at the workgroup's first tile and whenever the walk enters a new strip) -- every input byte is read once;
//   * the next tile's global loads are issued right after the current tile's registers have been stored to LDS and stay in flight under
//     the current tile's 72 MFMAs (register-staged: the loads land in the registers the stores have just released).
// Halo row r (0..3) of the tile in ring phase p (0 / 1, flips every step, 0 after a fresh load) lives in slot (2 p + r) & 3; the step code
// reaches the slots through four wave-uniform offsets (ONE copy of the MFMA body: see below).
static inline int wgrad_groups(int ntiles) {
    const int cap = g_conv_wgrad_groups > 0 ? g_conv_wgrad_groups : 512;      // 512 = two resident workgroups per CU; pa_debug_set(9, n): tests
    return ntiles < cap ? ntiles : cap;
}

__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ x, float* __restrict__ slab,
                                                               int Hi, int Wi, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 5;
    const int coh = wave & 1, cih = wave >> 1;
    const int ntx = Wi / TW, nty = Hi / WTH;
    // Staging map (chosen so that NOTHING per-chunk is worth hoisting out of the tile loop -- the first version of this kernel indexed
    // the 66-pixel halo rows linearly, hipcc kept ~40 loop-invariant index / address registers per thread alive across the MFMAs and
    // spilled 200 registers): thread = (pixel px0 = tid / 8, 16-byte chunk ch = tid % 8); a halo row's 64 body pixels are two chunks per
    // thread (px0, px0 + 32), the two edge columns of a row pair are one more chunk for threads 0..31.  The XOR swizzle of the LDS image
    // depends on pixel-column bits 1..3 only, so px0 + 32 (and the second dY row, + 64) swizzle like px0: one LDS base per thread.
    // (staged chunks as a plain array -- [0..3] body, [4] edge: a struct of uint4 members is copied through a stack slot that SROA does not always remove)
    const int px0 = tid >> 3, ch = tid & 7;
    const int e_rr = (tid >> 4) & 1, e_c = ((tid >> 3) & 1) ? TW + 1 : 0;        // edge chunk of threads 0..31: row of the pair, halo column 0 / 65
    const int xlds = (1 + px0) * 128 + ((ch ^ vsw(1 + px0)) << 4);               // body pixel px0 of a halo row (column 1 + px0)
    const int elds = e_c * 128 + ((ch ^ vsw(e_c)) << 4);
    const int ylds = W_XB + px0 * 128 + ((ch ^ vsw(px0)) << 4);
    auto tile_origin = [&](int tile, int& b, int& x0, int& y0) {
        const int strip = tile / nty, ty = tile - strip * nty;
        b = strip / ntx;
        x0 = (strip - b * ntx) * TW;
        y0 = ty * WTH;
    };
    // two halo rows (rbase = 0: the upper pair, only at a fresh start; 2: the pair every step fetches) of tile `tile` -> registers.
    // Straight-line: every lane loads from a clamped, valid address and a select zeroes what lies outside the image.
    auto load_rows = [&](uint4 (&r)[5], int tile, int rbase) {
        int b, x0, y0;
        tile_origin(tile, b, x0, y0);
        const bf16* img = x + (size_t)b * Hi * Wi * 64 + ch * 8;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int gy = y0 - 1 + rbase + rr;
            const bool rowok = gy >= 0 && gy < Hi;                               // wave-uniform
            const bf16* row = img + ((size_t)min(max(gy, 0), Hi - 1) * Wi + x0 + px0) * 64;
            const uint4 v0 = *reinterpret_cast<const uint4*>(row), v1 = *reinterpret_cast<const uint4*>(row + 32 * 64);
            r[2 * rr] = rowok ? v0 : zero4();
            r[2 * rr + 1] = rowok ? v1 : zero4();
        }
        const int gy = y0 - 1 + rbase + e_rr, gx = x0 - 1 + e_c;
        const bool ok = gy >= 0 && gy < Hi && gx >= 0 && gx < Wi;
        const uint4 ve = *reinterpret_cast<const uint4*>(img + ((size_t)min(max(gy, 0), Hi - 1) * Wi + min(max(gx, 0), Wi - 1)) * 64);
        r[4] = ok ? ve : zero4();
    };
    // registers -> LDS: halo row rbase + rr goes to ring slot (2 * phase + rbase + rr) & 3
    auto store_rows = [&](const uint4 (&r)[5], int rbase, int phase) {
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            unsigned char* d = smem + ((2 * phase + rbase + rr) & 3) * (HS * 128) + xlds;
            *reinterpret_cast<uint4*>(d) = r[2 * rr];
            *reinterpret_cast<uint4*>(d + 32 * 128) = r[2 * rr + 1];
        }
        if (tid < 32) *reinterpret_cast<uint4*>(smem + ((2 * phase + rbase + e_rr) & 3) * (HS * 128) + elds) = r[4];
    };
    uint4 xr[5];
    uint4 yr0, yr1, yr2, yr3;          // (named: the indexed form of these four ended up in a stack slot -- stored right behind their loads)
    auto load_dy = [&](int tile) {
        int b, x0, y0;
        tile_origin(tile, b, x0, y0);
        const bf16* dimg = dy + (((size_t)b * Hi + y0) * Wi + x0 + px0) * 64 + ch * 8;
        yr0 = *reinterpret_cast<const uint4*>(dimg);
        yr1 = *reinterpret_cast<const uint4*>(dimg + 32 * 64);
        yr2 = *reinterpret_cast<const uint4*>(dimg + (size_t)Wi * 64);
        yr3 = *reinterpret_cast<const uint4*>(dimg + ((size_t)Wi + 32) * 64);
    };
    auto store_dy = [&]() {
        *reinterpret_cast<uint4*>(smem + ylds) = yr0;
        *reinterpret_cast<uint4*>(smem + ylds + 32 * 128) = yr1;
        *reinterpret_cast<uint4*>(smem + ylds + 64 * 128) = yr2;
        *reinterpret_cast<uint4*>(smem + ylds + 96 * 128) = yr3;
    };

    // transposed-fragment addresses for 16-pixel step 0 of a row (steps / rows / ring slots are immediates)
    const int i16 = lane & 15, half = (lane >> 4) & 1;
    int ya[2], xa[3][2];
#pragma unroll
    for (int hi = 0; hi < 2; ++hi) {
        const int p = 4 * g + (i16 >> 2) + 8 * hi;
        const int sub = 2 * half + ((i16 & 3) >> 1);
        ya[hi] = W_XB + p * 128 + (((4 * coh + sub) ^ vsw(p)) << 4) + (i16 & 1) * 8;
#pragma unroll
        for (int dxi = 0; dxi < 3; ++dxi) {
            const int c = dxi + p;
            xa[dxi][hi] = c * 128 + (((4 * cih + sub) ^ vsw(c)) << 4) + (i16 & 1) * 8;
        }
    }

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // ONE copy of the 72-MFMA body: the ring phase enters through four wave-uniform slot offsets (halo row r of the tile lives at
    // srow[r]), added to the lane's fragment address per read.  (A first version instantiated the body per phase with immediate slot
    // offsets: hipcc then gave the nine accumulators different registers in the two copies and moved all 144 of them through scratch
    // between them -- 174 spilled registers, 921 us against the old kernel's 784.)
    int srow[4];
    auto step = [&](auto rr_c, auto m_c) {
        constexpr int rr = decltype(rr_c)::value, m = decltype(m_c)::value;
        constexpr int yo = (rr * TW + 16 * m) * 128;
        const bf16x8 a = tr_frag(smem + ya[0] + yo, smem + ya[1] + yo);
        auto one = [&](auto tap_c) {
            constexpr int tap = decltype(tap_c)::value;
            const int xo = srow[rr + tap / 3] + 16 * m * 128;
            const bf16x8 bb = tr_frag(smem + xa[tap % 3][0] + xo, smem + xa[tap % 3][1] + xo);
            acc[tap] = mfma(a, bb, acc[tap]);
        };
        one(IC<0>{}); one(IC<1>{}); one(IC<2>{}); one(IC<3>{}); one(IC<4>{}); one(IC<5>{}); one(IC<6>{}); one(IC<7>{}); one(IC<8>{});
        // fence per step: left alone, hipcc hoists the fragment reads of several steps above the first MFMA, takes every register that is
        // not an accumulator for them and parks the prefetched tile (xr / yr) in scratch -- a spill store waits for its load, which undoes
        // the prefetch.  One step's ten fragments in flight cover the LDS latency under nine MFMAs.
        __builtin_amdgcn_sched_barrier(0);
    };
    // this workgroup's run of the linear tile order (balanced to within one tile).  Every step's LOWER row pair and dY rows are
    // prefetched one tile ahead (xr / yr); the UPPER pair of a fresh start (the run's first tile, a new strip) is fetched on the spot --
    // an exposed round trip once or twice per workgroup, in exchange for 20 fewer staging registers across the MFMAs.
    const int t0 = (int)((int64_t)ntiles * blockIdx.x / gridDim.x), t1 = (int)((int64_t)ntiles * (blockIdx.x + 1) / gridDim.x);
    int phase = 0;
    if (t0 < t1) { load_rows(xr, t0, 2); load_dy(t0); }
#pragma clang loop unroll(disable)
    for (int tile = t0; tile < t1; ++tile) {
        const bool fresh = tile == t0 || tile % nty == 0;
        __syncthreads();                       // every wave has finished reading the slots / the dY image this tile overwrites
        if (fresh) {
            phase = 0;
            uint4 xq[5];
            load_rows(xq, tile, 0);
            store_rows(xq, 0, 0);
        }
        store_rows(xr, 2, phase);
        store_dy();
        if (tile + 1 < t1) { load_rows(xr, tile + 1, 2); load_dy(tile + 1); }      // travel under this tile's MFMAs
#pragma unroll
        for (int r = 0; r < 4; ++r) srow[r] = __builtin_amdgcn_readfirstlane(((2 * phase + r) & 3) * (HS * 128));
        __syncthreads();
        step(IC<0>{}, IC<0>{}); step(IC<0>{}, IC<1>{}); step(IC<0>{}, IC<2>{}); step(IC<0>{}, IC<3>{});
        step(IC<1>{}, IC<0>{}); step(IC<1>{}, IC<1>{}); step(IC<1>{}, IC<2>{}); step(IC<1>{}, IC<3>{});
        phase ^= 1;                            // (reset to 0 above when the next tile starts fresh)
    }
    float* o = slab + (size_t)blockIdx.x * W_SLAB;
    const int ci = cih * 32 + (lane & 31);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = coh * 32 + acc_row(r, lane);
            o[((size_t)co * 64 + ci) * 9 + tap] = acc[tap][r];
        }
}

static int launch_wgrad(const bf16* dy, const bf16* x, float* slab, int Bn, int Hi, int Wi, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    const int ntiles = Bn * (Hi / WTH) * (Wi / TW);
    PA_LAUNCH(conv3x3_wgrad_kernel, dim3(wgrad_groups(ntiles)), dim3(256), W_LDS, st, dy, x, slab, Hi, Wi, ntiles);
    return (int)hipGetLastError();
}

}   // namespace c64
