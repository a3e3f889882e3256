// Tile-level pieces of attn_tile.h for a head dim other than 64 (round 4: head_dim 80, ViT-H/14, BASELINE configs[4]) so that the
// generation-2 kernels (attn2.hip) can be instantiated per head dim.  TileOps<64> forwards to attn_tile.h's functions unchanged (the
// generation-2 / 3 code for every reference factory compiles to what it was); TileOps<80> keeps the first 64 columns of a 32-row tile in
// the very same 4 KB image and puts columns 64..79 into a 1 KB tail image behind it:
//     main  [32 rows][64 d]  XOR-swizzled 128-byte rows (attn_tile.h)              bytes [0, 4096)
//     tail  [32 rows][16 d]  32-byte rows, 16-byte chunk c of row r at c ^ (r >> 3 & 1)   bytes [4096, 5120)
// Contractions over d run five 16-deep k-steps (four from the main image, one from the tail); outputs with d rows have three 32-row
// blocks, the third one from the tail: its rows 16..31 (d = 80..95) do not exist -- the A fragments of those lanes are zero, the
// accumulator rows are never stored.
#pragma once
#include "attn_tile.h"

namespace atile {

template <int HD> struct TileOps;

template <> struct TileOps<64> {
    static constexpr int KS = 4, DB = 2, IMG_B = IMG, STG_B = IMG;
    typedef atile::Stager Stager;
    DEVI static bf16x8 rowfrag(const unsigned char* img, const LaneAddr& a, int s, int) { return atile::rowfrag(img, a, s); }
    DEVI static bf16x8 trfrag(const unsigned char* img, const LaneAddr& a, int db, int s, int) { return atile::trfrag(img, a, db, s); }
    // A fragment of a [64 d][ld] row-major global matrix for d-block db: rows db * 32 + (lane & 31)
    DEVI static bf16x8 gtfrag(const bf16* m, size_t ld, int db, int s, int lane) { return gfrag(m + (size_t)(db * 32 + (lane & 31)) * ld, s, lane >> 5); }
    DEVI static void stage_rows(unsigned char* stg, const f32x16 (&acc)[2], float mul, int lane) { atile::stage_rows(stg, acc, mul, lane); }
    DEVI static void write_rows(const unsigned char* stg, bf16* dst, size_t ld, int lane) { atile::write_rows(stg, dst, ld, lane); }
};

template <> struct TileOps<80> {
    static constexpr int KS = 5, DB = 3, IMG_B = IMG + 1024, STG_B = 32 * 160;
    DEVI static int tsw(int row) { return (row >> 3) & 1; }
    struct Stager {   // 32 x 80 bf16 = 320 chunks of 16 B: every thread one chunk of the main part, threads 0..63 one of the tail
        uint4 r, r2;
        DEVI void load(const bf16* src, size_t ld, int tid) {
            r = *reinterpret_cast<const uint4*>(src + (size_t)(tid >> 3) * ld + (tid & 7) * 8);
            r2 = zero4();          // (always defined: a partially initialised struct stays in scratch memory)
            if (tid < 64) r2 = *reinterpret_cast<const uint4*>(src + (size_t)(tid >> 1) * ld + 64 + (tid & 1) * 8);
        }
        DEVI void store(unsigned char* img, int tid) const {
            const int row = tid >> 3, c = tid & 7;
            *reinterpret_cast<uint4*>(img + row * 128 + ((c ^ vsw(row)) << 4)) = r;
            if (tid < 64) {
                const int rt = tid >> 1, ct = tid & 1;
                *reinterpret_cast<uint4*>(img + IMG + rt * 32 + ((ct ^ tsw(rt)) << 4)) = r2;
            }
        }
    };
    DEVI static bf16x8 rowfrag(const unsigned char* img, const LaneAddr& a, int s, int lane) {
        if (s < 4) return atile::rowfrag(img, a, s);
        const int row = lane & 31, g = lane >> 5;
        return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(img + IMG + row * 32 + ((g ^ tsw(row)) << 4)));
    }
    // transposed fragment (rows = d of block db, contraction over the tile's rows 16 s .. 16 s + 15).  Block 2: lane i of a 16-lane group
    // receives column 16 * half + i of the block (attn_tile.h); columns >= 16 do not exist: those lanes (half = 1) get zeros.
    DEVI static bf16x8 trfrag(const unsigned char* img, const LaneAddr& a, int db, int s, int lane) {
        if (db < 2) return atile::trfrag(img, a, db, s);
        const int i = lane & 15, half = (lane >> 4) & 1, g = lane >> 5;
        const int c0 = 4 * (i & 3);                                 // first of the lane's 4 columns
        const int r0 = 16 * s + 4 * g + (i >> 2), r1 = r0 + 8;
        const unsigned char* t = img + IMG;
        const u32x2 l = ldtr(t + r0 * 32 + (((c0 >> 3) ^ tsw(r0)) << 4) + (c0 & 7) * 2);
        const u32x2 h = ldtr(t + r1 * 32 + (((c0 >> 3) ^ tsw(r1)) << 4) + (c0 & 7) * 2);
        const u32x2 z = {0u, 0u};
        return __builtin_bit_cast(bf16x8, __builtin_shufflevector(half ? z : l, half ? z : h, 0, 1, 2, 3));
    }
    DEVI static bf16x8 gtfrag(const bf16* m, size_t ld, int db, int s, int lane) {
        const int row = db * 32 + (lane & 31);
        const bf16x8 v = gfrag(m + (size_t)min(row, 79) * ld, s, lane >> 5);
        return row < 80 ? v : __builtin_bit_cast(bf16x8, zero4());
    }
    // a wave's [d][row] accumulators (3 blocks of 32 d, the last one 16 live rows) -> bf16 rows [32][80] in LDS (160-byte rows) -> global
    DEVI static void stage_rows(unsigned char* stg, const f32x16 (&acc)[3], float mul, int lane) {
        const int g = lane >> 5;
#pragma unroll
        for (int db = 0; db < 3; ++db)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int d0 = db * 32 + 8 * rg + 4 * g;
                if (d0 < 80)
                    *reinterpret_cast<uint2*>(stg + (lane & 31) * 160 + d0 * 2) =
                        make_uint2(pack_bf16x2(acc[db][rg * 4] * mul, acc[db][rg * 4 + 1] * mul), pack_bf16x2(acc[db][rg * 4 + 2] * mul, acc[db][rg * 4 + 3] * mul));
            }
    }
    DEVI static void write_rows(const unsigned char* stg, bf16* dst, size_t ld, int lane) {
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int c = lane + 64 * i, row = c / 10, ch = c - row * 10;
            *reinterpret_cast<uint4*>(dst + (size_t)row * ld + ch * 8) = *reinterpret_cast<const uint4*>(stg + row * 160 + ch * 16);
        }
    }
};

}   // namespace atile
