// Shared pieces of the fused attention kernels (forward, backward-dQ, backward-dKV).
//
// Math (per sample b', head h; Painter/models_painter.py:73-89, util/vitdet_utils.py:96-125):
//   S[q,k] = scale * q.k + q.rel_pos_h[qh - kh + Hp-1] + q.rel_pos_w[qw - kw + Wp-1]   (bias uses UNSCALED q)
//   P = softmax_k(S),  O = P V
// Layout facts used everywhere:
//   * token l = h * Wp + w; qkv buffer row = [q(h0..), k(h0..), v(h0..)] with head stride HD.  The generation-1 kernels of
//     attn_fwd.hip / attn_bwd.hip are templated on HD (64: every reference factory; 80: ViT-H/14, BASELINE configs[4]); the bf16
//     generations 2 and 3 (attn2.hip, attn3.hip) are HD = 64 only (ATT_HD).
//   * "r-space": G[q][r] = q . Rcat[r],  Rcat = [rel_pos_h (2Hp-1 rows); rel_pos_w (2Wp-1 rows); 0 pad] (NRP rows).
//   * "k-space" table per query: tab[kh] = G[q][qh + Hp-1 - kh], tab[Hp + kw] = G[q][2Hp-1 + qw + Wp-1 - kw];
//     TS = Hp + Wp floats per query.  Tables hold bias * log2(e) (softmax runs in the exp2 domain).
//   * A run of 4 consecutive keys starting at a multiple of 4 never crosses a key row (Wp % 4 == 0).
#pragma once
#include "common.h"

#define ATT_HD 64
#define LOG2E_F 1.4426950408889634f
#define LN2_F 0.6931471805599453f

// LDS images of one 32-row tile of a [rows][HD] operand:
//   row image  [32 rows][HD]  (contraction over HD: Q.K^T, dO.V^T)           KB bytes
//   transposed [HDP d][32 rows], HDP = HD rounded up to 32 (contraction over the 32 rows: P.V, dS^T.Q ...)   VB bytes;
//              rows d >= HD are zero padding (the 32-row MFMA output blocks are whole), written once per kernel by zero_pad().
// HD = 64 keeps the XOR-swizzled 128 / 256-byte rows of common.h; any other HD pads each row by one 16-byte chunk instead.
template <typename T, int HD> struct KvTile {
    static constexpr int ES = sizeof(T);
    static constexpr int KS = HD / 16;                  // 16-deep MFMA k-steps of a contraction over HD
    static constexpr int DB = (HD + 31) / 32;           // 32-row blocks of an output with HD rows
    static constexpr int HDP = DB * 32;
    static constexpr int RS = HD == 64 ? HD * ES : HD * ES + 16;
    static constexpr int KB = 32 * RS;
    static constexpr int VB = HDP * 32 * ES;
    static_assert(HD % 16 == 0 && HD >= 16 && HD <= 128, "head_dim must be a multiple of 16");
    DEVI static int k_off(int row, int slot) {           // 16-byte slot `slot` of row `row`
        if constexpr (HD == 64) return ES == 2 ? lds128(row, slot) : lds256(row, slot);
        else return row * RS + slot * 16;
    }
    DEVI static int vt_off(int d, int unit) {            // [d][4-row unit]: 8 units of 8 B (bf16) / 16 B (float)
        return ES == 2 ? lds64(d, unit) : lds128(d, unit);
    }
    // zero the padding rows [HD, HDP) of a transposed image (NT threads)
    DEVI static void zero_pad(unsigned char* vt, int tid, int NT) {
        if constexpr (HDP != HD) {
            for (int i = tid; i < (HDP - HD) * 32 * ES / 16; i += NT) *reinterpret_cast<uint4*>(vt + HD * 32 * ES + i * 16) = zero4();
        }
    }
};

// row-major [32][HD] operand fragment for k-step s (contraction over the HD-wide axis)
template <typename T, int HD> DEVI void load_rowfrag(Frag<T>& f, const unsigned char* tile, int row, int s, int g) {
    if constexpr (sizeof(T) == 2) {
        f.set(*reinterpret_cast<const uint4*>(tile + KvTile<T, HD>::k_off(row, s * 2 + g)));
    } else {
        f.set(*reinterpret_cast<const uint4*>(tile + KvTile<T, HD>::k_off(row, (s * 2 + g) * 2)),
              *reinterpret_cast<const uint4*>(tile + KvTile<T, HD>::k_off(row, (s * 2 + g) * 2 + 1)));
    }
}
// transposed [HDP][32] operand fragment for k-step s (contraction over the 32-wide axis), slot order
// (g,t) <-> index 16 s + 4 g + (t & 3) + 8 (t >> 2): exactly the order the MFMA D layout hands out.
template <typename T, int HD> DEVI void load_trfrag(Frag<T>& f, const unsigned char* tile, int row, int s, int g) {
    if constexpr (sizeof(T) == 2) {
        const uint2 a = *reinterpret_cast<const uint2*>(tile + KvTile<T, HD>::vt_off(row, 4 * s + g));
        const uint2 b = *reinterpret_cast<const uint2*>(tile + KvTile<T, HD>::vt_off(row, 4 * s + g + 2));
        f.set(make_uint4(a.x, a.y, b.x, b.y));
    } else {
        f.set(*reinterpret_cast<const uint4*>(tile + KvTile<T, HD>::vt_off(row, 4 * s + g)),
              *reinterpret_cast<const uint4*>(tile + KvTile<T, HD>::vt_off(row, 4 * s + g + 2)));
    }
}
// fragment straight from a global row (HD contiguous T at p): k-step s, group g
template <typename T> DEVI void load_gfrag(Frag<T>& f, const T* p, int s, int g) {
    if constexpr (sizeof(T) == 2) {
        f.set(*reinterpret_cast<const uint4*>(p + 16 * s + 8 * g));
    } else {
        f.set(*reinterpret_cast<const uint4*>(p + 16 * s + 8 * g), *reinterpret_cast<const uint4*>(p + 16 * s + 8 * g + 4));
    }
}
// pack 8 accumulator values (regs 8s .. 8s+7 of a D tile) into an operand fragment
template <typename T> DEVI void pack_frag(Frag<T>& f, const float* v) {
    if constexpr (sizeof(T) == 2) {
        f.set(make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])));
    } else {
#pragma unroll
        for (int t = 0; t < 8; ++t) f.v[t] = v[t];
    }
}

// cooperative staging of a [32 rows][HD] row-major tile (rows at src + row * ld) into the row image,
// and of its transpose into the transposed image.  NT threads.
template <typename T, int NT, int HD> struct RowStage {
    static constexpr int SL = HD * sizeof(T) / 16;          // 16-B chunks per row
    static constexpr int NCH = 32 * SL;
    static constexpr int CK = (NCH + NT - 1) / NT;
    uint4 r[CK];
    DEVI void load(const T* src, size_t ld, int tid) {
#pragma unroll
        for (int i = 0; i < CK; ++i) {
            const int c = tid + NT * i;
            if (c < NCH) r[i] = *reinterpret_cast<const uint4*>(src + (size_t)(c / SL) * ld + (c % SL) * TT<T>::EPC);
        }
    }
    DEVI void store(unsigned char* tile, int tid) const {
#pragma unroll
        for (int i = 0; i < CK; ++i) {
            const int c = tid + NT * i;
            if (c < NCH) *reinterpret_cast<uint4*>(tile + KvTile<T, HD>::k_off(c / SL, c % SL)) = r[i];
        }
    }
};
template <typename T, int HD> struct TrStage {      // threads 0 .. 2*HD-1: one 4(row) x 4(col) block each (NT >= 2*HD)
    typedef typename TT<T>::Vec4 Vec4;
    static constexpr int CBN = HD / 4;              // column blocks
    Vec4 r[4];
    DEVI void load(const T* src, size_t ld, int tid) {
        if (tid < 8 * CBN) {
            const int cb = tid % CBN, rb = tid / CBN;
#pragma unroll
            for (int i = 0; i < 4; ++i) r[i] = *reinterpret_cast<const Vec4*>(src + (size_t)(rb * 4 + i) * ld + cb * 4);
        }
    }
    DEVI void store(unsigned char* tile, int tid) const {
        if (tid < 8 * CBN) {
            const int cb = tid % CBN, rb = tid / CBN;
            Vec4 o[4];
            transpose4x4(r, o);
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<Vec4*>(tile + KvTile<T, HD>::vt_off(cb * 4 + j, rb)) = o[j];
        }
    }
};

// G^T = Rcat . Q^T for this lane's query row, scattered into the k-space table (times log2 e).
template <typename T, int HD>
DEVI void build_bias_table(float* tab, const T* rcat, int NRP, const Frag<T> (&qf)[HD / 16], int qh, int qw, int Hp, int Wp, int lane) {
    const int g = lane >> 5;
    for (int rbk = 0; rbk < NRP / 32; ++rbk) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const T* rp = rcat + (size_t)(rbk * 32 + (lane & 31)) * HD;
#pragma unroll
        for (int s = 0; s < HD / 16; ++s) {
            Frag<T> a;
            load_gfrag<T>(a, rp, s, g);
            mma(acc, a, qf[s]);
        }
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int r = rbk * 32 + acc_row(reg, lane);
            const float v = acc[reg] * LOG2E_F;
            if (r < 2 * Hp - 1) {
                const int kh = qh + Hp - 1 - r;
                if (kh >= 0 && kh < Hp) tab[kh] = v;
            } else {
                const int rr = r - (2 * Hp - 1);
                const int kw = qw + Wp - 1 - rr;
                if (rr < 2 * Wp - 1 && kw >= 0 && kw < Wp) tab[Hp + kw] = v;
            }
        }
    }
}
