// Tile-level pieces shared by the bf16 attention kernels (attn2.hip: any key-row width, bias through k-space tables;
// attn3.hip: key rows of 28, bias on the matrix pipe).  Math: Painter/models_painter.py:76-86, util/vitdet_utils.py:96-125.
//
// Work split of every kernel: workgroup = 4 waves, wave = 32 rows of the stationary axis (queries, or keys in dKV), lane = one row end
// to end; 32-row tiles of the other axis stream through LDS.  One LDS image per streamed tile: natural [row][64 d] order, XOR-swizzled
// so that BOTH the row-fragment ds_read_b128 (contraction over d) and the transposing ds_read_b64_tr_b16 (contraction over the tile's
// rows) are bank-conflict free -- no transposed copies, no register transposes.
//
// fp32 VALU arithmetic here is written one scalar operation at a time on purpose: packed fp32 (v_pk_*_f32, whether hand-written with
// ext_vector_type(2) floats or produced by the SLP vectoriser) is both slower beside MFMAs on gfx950 and the instruction class that
// mis-computed in the LayerNorm backward when another kernel's MFMA workgroups shared the CU.
#pragma once
#include "attn_common.h"

namespace atile {

constexpr int NW = 4, NT = 256, ROWS = 128, IMG = 4096, STAGE_QK = 2 * IMG;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((ext_vector_type(8))) float f32x8;

// 1-D grid of (row blocks per head) x (batch * heads) workgroups.  Workgroup n runs on XCD n % 8 (MI355X dispatch order), and each
// XCD has its own 4 MB L2: give every XCD a contiguous run of (head, row block) pairs, row block fastest, so that all row blocks
// of a head stream the head's K/V (or Q/dO/aux) tiles through ONE L2 instead of eight (7-8 heads x 0.4 MB live per XCD).
// PA_ATTN_XCD=0 (diagnostics) keeps the plain order.
// xcd_map bit 1 (generation 3, "light last"): when L / 32 is not a multiple of the NW row blocks a workgroup takes, the last workgroup
// of every head has idle waves (1568 tokens: 12 full workgroups + one with a single live wave, which still walks the whole key loop).
// The full workgroups are numbered first -- 12 x 128 heads = 1536 = exactly three rounds of the 512 resident workgroups at the
// ViT-L B = 8 shape -- and the light ones take the last block indices, i.e. they are dispatched last and share the final partial round
// among themselves instead of holding a slot beside full workgroups.  Both sets keep the XCD-contiguous head order.
DEVI int xcd_run(int n, int total) {
    const int xq = total >> 3, xr = total & 7, xcd = n & 7;
    return (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (n >> 3);
}
DEVI void wg_coords(int nblk, int xcd_map, int& blk, int& bh, int nfull = 0) {
    const int n = blockIdx.x, total = gridDim.x;
    if ((xcd_map & 2) && nfull > 0 && nfull < nblk) {
        const int nbh = total / nblk, heavy = nfull * nbh;
        if (n < heavy) {
            const int v = (xcd_map & 1) ? xcd_run(n, heavy) : n;
            bh = v / nfull;
            blk = v - bh * nfull;
        } else {
            // only nblk == nfull + 1 reaches here (one light workgroup per head)
            const int m = n - heavy, per = nblk - nfull, lt = nbh * per;
            const int v = (xcd_map & 1) ? xcd_run(m, lt) : m;
            bh = v / per;
            blk = nfull + (v - bh * per);
        }
        return;
    }
    const int v = (xcd_map & 1) ? xcd_run(n, total) : n;
    bh = v / nblk;
    blk = v - bh * nblk;
}

// ---- tile image: 32 rows x 128 B, 16-B chunk index XORed with a bijection of row bits 1..3
DEVI int vsw(int row) { return (((row >> 1) & 1) << 2) | (((row >> 2) & 1) << 1) | ((row >> 3) & 1); }

struct Stager {   // one 32 x 64 bf16 tile, 256 threads, one 16-B chunk each
    uint4 r;
    DEVI void load(const bf16* src, size_t ld, int tid) { r = *reinterpret_cast<const uint4*>(src + (size_t)(tid >> 3) * ld + (tid & 7) * 8); }
    DEVI void store(unsigned char* img, int tid) const {
        const int row = tid >> 3, c = tid & 7;
        *reinterpret_cast<uint4*>(img + row * 128 + ((c ^ vsw(row)) << 4)) = r;
    }
};

// per-lane address pieces, computed once
struct LaneAddr {
    int rowbase, t;          // row fragment: byte = rowbase + (((2 s) ^ t) << 4)
    int tr[2][2];            // transposed fragment: [dblk][lo/hi] byte offset for k-step 0; k-step 1 = + 2048
    DEVI void init(int lane) {
        const int row = lane & 31, g = lane >> 5;
        rowbase = row * 128;
        t = vsw(row) ^ g;
        const int i = lane & 15, half = (lane >> 4) & 1;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int hi = 0; hi < 2; ++hi) {
                const int r = 4 * g + (i >> 2) + 8 * hi;
                const int chunk = 4 * db + 2 * half + ((i & 3) >> 1);
                tr[db][hi] = r * 128 + ((chunk ^ vsw(r)) << 4) + (i & 1) * 8;
            }
    }
};
// A operand, rows = tile rows, contraction over d (k-step s of 16)
DEVI bf16x8 rowfrag(const unsigned char* img, const LaneAddr& a, int s) {
    return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(img + a.rowbase + (((2 * s) ^ a.t) << 4)));
}
// transposing 8-byte LDS read: within each group of 16 lanes, lane i receives element (i & 3) of the four lanes 4 e + (i >> 2), e = 0..3
DEVI u32x2 ldtr(const unsigned char* p) {
    return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p));
}
// A operand, rows = d (block db of 32), contraction over the tile's rows: slot t <-> row 16 s + 4 g + (t & 3) + 8 (t >> 2),
// the order in which the MFMA D layout hands a lane its values (so D registers pack straight into the B operand)
DEVI bf16x8 trfrag(const unsigned char* img, const LaneAddr& a, int db, int s) {
    const u32x2 l = ldtr(img + a.tr[db][0] + s * 2048), h = ldtr(img + a.tr[db][1] + s * 2048);
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(l, h, 0, 1, 2, 3));
}
DEVI bf16x8 gfrag(const bf16* p, int s, int g) { return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(p + 16 * s + 8 * g)); }
DEVI bf16x8 packfrag(const float* v) {      // one v_cvt_pk_bf16_f32 per pair
    const f32x8 f = {v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]};
    return __builtin_convertvector(f, bf16x8);
}
// exchange with the lane 32 away (the other half of this row's key runs): VALU permlane, no LDS round trip
DEVI float xor32(float v) {
    const uint32_t u = __builtin_bit_cast(uint32_t, v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, (threadIdx.x & 32) ? r[0] : r[1]);
}
DEVI float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
DEVI float max16(const float* p) {
    const float a = max3f(p[0], p[1], p[2]), b = max3f(p[3], p[4], p[5]), c = max3f(p[6], p[7], p[8]), d = max3f(p[9], p[10], p[11]),
                e = max3f(p[12], p[13], p[14]);
    return max3f(max3f(a, b, c), max3f(d, e, p[15]), -INFINITY);
}
DEVI float sum16(const float* p) {      // four independent chains, scalar adds
    float a = p[0] + p[4], b = p[1] + p[5], c = p[2] + p[6], d = p[3] + p[7];
    a += p[8]; b += p[9]; c += p[10]; d += p[11];
    a += p[12]; b += p[13]; c += p[14]; d += p[15];
    return (a + b) + (c + d);
}
DEVI f32x16 mfma(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }

// stage a wave's [d][row] accumulators (2 blocks of 32 d) as bf16 rows in LDS, then write whole 128-B rows
DEVI void stage_rows(unsigned char* stg, const f32x16 (&acc)[2], float mul, int lane) {
    const int g = lane >> 5;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int d0 = db * 32 + 8 * rg + 4 * g;
            *reinterpret_cast<uint2*>(stg + (lane & 31) * 128 + d0 * 2) =
                make_uint2(pack_bf16x2(acc[db][rg * 4] * mul, acc[db][rg * 4 + 1] * mul), pack_bf16x2(acc[db][rg * 4 + 2] * mul, acc[db][rg * 4 + 3] * mul));
        }
}
DEVI void write_rows(const unsigned char* stg, bf16* dst, size_t ld, int lane) {   // dst = row 0 of the wave's 32 rows
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i, row = c >> 3, ch = c & 7;
        *reinterpret_cast<uint4*>(dst + (size_t)row * ld + ch * 8) = *reinterpret_cast<const uint4*>(stg + row * 128 + ch * 16);
    }
}

static inline int set_smem(const void* kern, bool& done) {
    if (!done) {
        hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        done = true;
    }
    return 0;
}

}   // namespace atile
