// Shared pieces of the fused attention kernels (forward, backward-dQ, backward-dKV).
//
// Math (per sample b', head h; Painter/models_painter.py:73-89, util/vitdet_utils.py:96-125):
//   S[q,k] = scale * q.k + q.rel_pos_h[qh - kh + Hp-1] + q.rel_pos_w[qw - kw + Wp-1]   (bias uses UNSCALED q)
//   P = softmax_k(S),  O = P V
// Layout facts used everywhere:
//   * token l = h * Wp + w; qkv buffer row = [q(h0..), k(h0..), v(h0..)] with head stride 64 (HD = 64).
//   * "r-space": G[q][r] = q . Rcat[r],  Rcat = [rel_pos_h (2Hp-1 rows); rel_pos_w (2Wp-1 rows); 0 pad] (NRP rows).
//   * "k-space" table per query: tab[kh] = G[q][qh + Hp-1 - kh], tab[Hp + kw] = G[q][2Hp-1 + qw + Wp-1 - kw];
//     TS = Hp + Wp floats per query.  Tables hold bias * log2(e) (softmax runs in the exp2 domain).
//   * A run of 4 consecutive keys starting at a multiple of 4 never crosses a key row (Wp % 4 == 0).
#pragma once
#include "common.h"

#define ATT_HD 64
#define LOG2E_F 1.4426950408889634f
#define LN2_F 0.6931471805599453f

template <typename T> struct KvTile;   // LDS images of one 32-key K / V tile
template <> struct KvTile<bf16> {
    static constexpr int BYTES = 32 * 64 * 2;
    DEVI static int k_off(int key, int slot) { return lds128(key, slot); }       // K[key][d]: 8 slots of 8 d
    DEVI static int vt_off(int d, int unit) { return lds64(d, unit); }           // Vt[d][key]: 8 units of 4 keys (8 B)
};
template <> struct KvTile<float> {
    static constexpr int BYTES = 32 * 64 * 4;
    DEVI static int k_off(int key, int slot) { return lds256(key, slot); }       // 16 slots of 4 d
    DEVI static int vt_off(int d, int unit) { return lds128(d, unit); }          // 8 units of 4 keys (16 B)
};

// row-major [32][64] operand fragment for k-step s (contraction over the 64-wide axis)
template <typename T> DEVI void load_rowfrag(Frag<T>& f, const unsigned char* tile, int row, int s, int g) {
    if constexpr (sizeof(T) == 2) {
        f.set(*reinterpret_cast<const uint4*>(tile + KvTile<T>::k_off(row, s * 2 + g)));
    } else {
        f.set(*reinterpret_cast<const uint4*>(tile + KvTile<T>::k_off(row, (s * 2 + g) * 2)),
              *reinterpret_cast<const uint4*>(tile + KvTile<T>::k_off(row, (s * 2 + g) * 2 + 1)));
    }
}
// transposed [64][32] operand fragment for k-step s (contraction over the 32-wide axis), slot order
// (g,t) <-> index 16 s + 4 g + (t & 3) + 8 (t >> 2): exactly the order the MFMA D layout hands out.
template <typename T> DEVI void load_trfrag(Frag<T>& f, const unsigned char* tile, int row, int s, int g) {
    if constexpr (sizeof(T) == 2) {
        const uint2 a = *reinterpret_cast<const uint2*>(tile + KvTile<T>::vt_off(row, 4 * s + g));
        const uint2 b = *reinterpret_cast<const uint2*>(tile + KvTile<T>::vt_off(row, 4 * s + g + 2));
        f.set(make_uint4(a.x, a.y, b.x, b.y));
    } else {
        f.set(*reinterpret_cast<const uint4*>(tile + KvTile<T>::vt_off(row, 4 * s + g)),
              *reinterpret_cast<const uint4*>(tile + KvTile<T>::vt_off(row, 4 * s + g + 2)));
    }
}
// fragment straight from a global row (64 contiguous T at p): k-step s, group g
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

// cooperative staging of a [32 rows][64] row-major tile (rows at src + row * ld) into the K image,
// and of its transpose into the Vt image.  NT threads.
template <typename T, int NT> struct RowStage {
    static constexpr int SL = 64 * sizeof(T) / 16;          // 16-B chunks per row
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
            if (c < NCH) *reinterpret_cast<uint4*>(tile + KvTile<T>::k_off(c / SL, c % SL)) = r[i];
        }
    }
};
template <typename T> struct TrStage {      // threads 0..127: one 4(row) x 4(col) block each
    typedef typename TT<T>::Vec4 Vec4;
    Vec4 r[4];
    DEVI void load(const T* src, size_t ld, int tid) {
        if (tid < 128) {
            const int cb = tid & 15, rb = tid >> 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) r[i] = *reinterpret_cast<const Vec4*>(src + (size_t)(rb * 4 + i) * ld + cb * 4);
        }
    }
    DEVI void store(unsigned char* tile, int tid) const {
        if (tid < 128) {
            const int cb = tid & 15, rb = tid >> 4;
            Vec4 o[4];
            transpose4x4(r, o);
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<Vec4*>(tile + KvTile<T>::vt_off(cb * 4 + j, rb)) = o[j];
        }
    }
};

// G^T = Rcat . Q^T for this lane's query row, scattered into the k-space table (times log2 e).
template <typename T>
DEVI void build_bias_table(float* tab, const T* rcat, int NRP, const Frag<T> (&qf)[4], int qh, int qw, int Hp, int Wp, int lane) {
    const int g = lane >> 5;
    for (int rbk = 0; rbk < NRP / 32; ++rbk) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const T* rp = rcat + (size_t)(rbk * 32 + (lane & 31)) * ATT_HD;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
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
