// split_core.h -- shared machinery of the opt-in "bf16x3" kernels (tcn_split.hip, gcn_split.hip): fp32 operands as three
// bf16 pieces, K-contiguous operand images, weight / activation staging, the 6-product MFMA step and the store epilogue.
#pragma once
#include "mfma_core.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

static constexpr int KS = 16;          // channels per K step / chunk of the split kernels

// x -> (h, m, l) for 8 values: v_cvt_pk_bf16_f32 (round to nearest even) for the pieces, exact fp32 subtractions
__device__ __forceinline__ void split8(const float (&x)[8], bf16x8 &h, bf16x8 &m, bf16x8 &l) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 hb = (__bf16)x[j];
        const float r1 = x[j] - (float)hb;
        const __bf16 mb = (__bf16)r1;
        const float r2 = r1 - (float)mb;
        h[j] = hb; m[j] = mb; l[j] = (__bf16)r2;
    }
}

// ---- stage kernel: 512 threads = 8 waves (two per SIMD), ONE workgroup per CU, tile MT x NT with MT * NT = 32768 (128 x 256
// or 64 x 512; every wave a 64 x 64 block as in tcn.hip).  Compared with a 4-wave 128 x 128 tile at two workgroups per CU
// (the first form, measured 0.75-1.24x the exact-fp32 kernel: two barriers and a full weight stage per 72 MFMAs) a staged
// weight vector feeds twice the MFMAs, and the weight stages ping-pong between two LDS buffers so that a stage costs ONE
// barrier: the next stage's weights are committed in front of this stage's MFMAs, the stage after that is in flight in
// registers.  The activation tile is single-buffered (two buffers do not fit): one extra barrier pair per 16-channel chunk.
static constexpr int NTH2 = 512;
static constexpr int TG = 3, NSTAGE = 3;     // 9 taps = 3 weight stages of 3 taps

template <int MT>
struct WSplitStage {
    static constexpr int NV = TG * 6 * MT;                  // 16-byte vectors of a stage (3 tap slots)
    static constexpr int WB = (NV + NTH2 - 1) / NTH2;
    u32x4 v[WB];
    // vector e = u * 512 + tid of the stage (surplus threads re-stage the last one); offsets are recomputed per use
    // (MT is a power of two: a shift and a mask) instead of being held in registers
    __device__ __forceinline__ void issue(const u32x4 *__restrict__ base, int Mpad, int tid) {
#pragma unroll
        for (int u = 0; u < WB; ++u) {
            const int e = min(u * NTH2 + tid, NV - 1);
            v[u] = base[(e / MT) * Mpad + (e % MT)];
        }
    }
    __device__ __forceinline__ void commit(u32x4 *__restrict__ Wl, int tid) const {
#pragma unroll
        for (int u = 0; u < WB; ++u) Wl[min(u * NTH2 + tid, NV - 1)] = v[u];
    }
};

// activations of one 16-channel chunk: wave w stages k-half h = w & 1 of the position sweeps (w >> 1) + 4 i (64 tile
// positions each): per sweep a lane loads the 8 channels of its position, unconditionally (clamped address + select, see
// BStage in mfma_core.h), and at commit time splits them into the three piece fragments.  setup() maps tile positions
// to source positions for a kind (frame de-interleave, zero padding outside [0, T)).
template <int NS4>
struct BSplitStage {
    unsigned goff[NS4], loff[NS4], valid;
    int h;
    float v[NS4][8];
    // tile position j -> (class, frame in class, joint) -> source position; classes: ncls sets of (dt + ntap[rho]) frames,
    // source frame of tile frame jf of class rho = fbase + rho + fstep * (jf - first frame of the class)
    __device__ __forceinline__ void setup(int fbase, int fstep, int dt, int ncls, const int (&ntap)[4], int T, int V, unsigned vmagic,
                                          int lane, int wave) {
        h = wave & 1;
        valid = 0;
        int total = 0;
        for (int r = 0; r < ncls; ++r) total += dt + ntap[r];
#pragma unroll
        for (int i = 0; i < NS4; ++i) {
            const int j = min(((wave >> 1) + 4 * i) * 64 + lane, total * V - 1);        // tile position
            const int jf = div_magic(j, vmagic);
            int cls = 0, fb = 0, acc = 0;
            for (int r = 0; r < ncls; ++r) {
                if (jf >= acc) { cls = r; fb = acc; }
                acc += dt + ntap[r];
            }
            const int f = fbase + cls + fstep * (jf - fb);
            goff[i] = (unsigned)(min(max(f, 0), T - 1) * V + (j - jf * V));
            loff[i] = (unsigned)j;
            valid |= (f >= 0 && f < T) ? (1u << i) : 0u;
        }
    }
    __device__ __forceinline__ void issue(const float *__restrict__ seg_base, int C, int64_t cs, int c0) {
        // (the mask is formed per channel row and applied with a plain select: a short-circuit `c < C && bit` makes hipcc
        // branch around every load and wait for each one separately)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = c0 + 8 * h + j;
            const float *src = seg_base + (int64_t)min(c, C - 1) * cs;
            const unsigned m = c < C ? valid : 0u;
#pragma unroll
            for (int i = 0; i < NS4; ++i) {
                const float x = src[goff[i]];
                v[i][j] = ((m >> i) & 1u) ? x : 0.f;
            }
        }
    }
    __device__ __forceinline__ void commit(u32x4 *__restrict__ Bl, int ldb) const {
#pragma unroll
        for (int i = 0; i < NS4; ++i) {
            bf16x8 ph, pm, pl;
            split8(v[i], ph, pm, pl);
            u32x4 *dst = Bl + h * ldb + loff[i];
            dst[0] = __builtin_bit_cast(u32x4, ph);
            dst[2 * ldb] = __builtin_bit_cast(u32x4, pm);
            dst[4 * ldb] = __builtin_bit_cast(u32x4, pl);
        }
    }
    // the same in two steps: the split (vector ALU work) issued in front of a stage's MFMAs, where it runs beside the matrix
    // pipe, and the LDS writes behind the barrier that frees the tile
    u32x4 pk[NS4][3];
    __device__ __forceinline__ void presplit() {
#pragma unroll
        for (int i = 0; i < NS4; ++i) {
            bf16x8 ph, pm, pl;
            split8(v[i], ph, pm, pl);
            pk[i][0] = __builtin_bit_cast(u32x4, ph);
            pk[i][1] = __builtin_bit_cast(u32x4, pm);
            pk[i][2] = __builtin_bit_cast(u32x4, pl);
        }
    }
    __device__ __forceinline__ void commit_pk(u32x4 *__restrict__ Bl, int ldb) const {
#pragma unroll
        for (int i = 0; i < NS4; ++i) {
            u32x4 *dst = Bl + h * ldb + loff[i];
            dst[0] = pk[i][0];
            dst[2 * ldb] = pk[i][1];
            dst[4 * ldb] = pk[i][2];
        }
    }
};

// NTAP taps of the staged weights against the activation tile: per tap 12 ds_read_b128 (3 pieces x (2 row + 2 column
// blocks)) and 24 MFMAs (6 piece products x 2 x 2 blocks), small products first.  Unrolled over the taps of a stage so
// that the scheduler can run a tap's fragment reads under the previous tap's MFMAs.
template <int MT, int NTAP>
__device__ __forceinline__ void mfma_split_taps(const u32x4 *__restrict__ Wl, const u32x4 *__restrict__ Bl, int ldb, const int *toff,
                                                int offA, int off0, int off1, int kh, f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int t = 0; t < NTAP; ++t) {
        const u32x4 *wr = Wl + (t * 6 + kh) * MT + offA;
        const u32x4 *br = Bl + kh * ldb + toff[t];
        bf16x8 a[3][2], b[3][2];
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) {
            a[pc][0] = __builtin_bit_cast(bf16x8, wr[pc * 2 * MT]);
            a[pc][1] = __builtin_bit_cast(bf16x8, wr[pc * 2 * MT + 32]);
            b[pc][0] = __builtin_bit_cast(bf16x8, br[pc * 2 * ldb + off0]);
            b[pc][1] = __builtin_bit_cast(bf16x8, br[pc * 2 * ldb + off1]);
        }
        constexpr int PA[6] = {0, 2, 1, 1, 0, 0}, PB[6] = {2, 0, 1, 0, 1, 0};       // hl, lh, mm, mh, hm, hh
#pragma unroll
        for (int t6 = 0; t6 < 6; ++t6)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[t6]][mi], b[PB[t6]][ni], acc[mi][ni], 0, 0, 0);
    }
}

// Epilogue shared by the split kernels: + bias (+ identity residual), ReLU, stores -- the scheme of tcn_stage_kernel
// (scalar row bases + 32-bit lane offsets on full tiles, permlane32_swap for 256-B row segments), operands loaded one
// 32-row half at a time.  Output position q of row co at out[co * Q + q]; identity residual of (co, q = t V + v) at
// xres[co * Tres * V + (t * stride + res_off) * V + v].
struct SplitEpi {
    const float *bias, *rseg;      // folded bias; residual segment base (or any valid pointer when !ident)
    float *oseg;                   // output segment base
    int Cout, Tres, V, Q, stride, res_off, relu, ident, fast_epi;
    unsigned vmagic;
};

template <int MT>
__device__ __forceinline__ void split_epilogue(const SplitEpi &p, f32x16 (&acc)[2][2], int m0, int wm, int wn, int q0, int qend,
                                               int lane) {
    const int l31 = lane & 31, kh = lane >> 5;
    const int V = p.V, Q = p.Q;
    float *oseg = p.oseg;
    const float *rseg = p.rseg;
    const int64_t rcs = (int64_t)p.Tres * V;
    const bool ident = p.ident;
    const int rbase = m0 + wm * 64;
    const bool full = p.fast_epi && m0 + MT <= p.Cout;
    const unsigned kh4 = 4u * (unsigned)kh;
    const int qb = q0 + wn * 64 + lane;
    const bool qv = qb < qend;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        float bv[16], rv[2][16];
        if (full) {
#pragma unroll
            for (int g = 0; g < 16; ++g) bv[g] = ld_lane(p.bias + (rbase + mi * 32 + (g & 3) + 8 * (g >> 2)), kh4 * 4u);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int qc = min(q0 + wn * 64 + ni * 32 + l31, qend - 1);
                const int t = div_magic(qc, p.vmagic);
                const unsigned qres = ident ? 4u * (kh4 * (unsigned)rcs + (unsigned)((t * p.stride + p.res_off) * V + (qc - t * V))) : 0u;
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const float *rrow = rseg + (int64_t)(rbase + mi * 32 + (g & 3) + 8 * (g >> 2)) * rcs;
                    rv[ni][g] = ident ? ld_lane(rrow, qres) : 0.f;
                }
            }
        } else {
#pragma unroll
            for (int g = 0; g < 16; ++g) bv[g] = p.bias[rbase + mi * 32 + 4 * kh + (g & 3) + 8 * (g >> 2)];
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int qc = min(q0 + wn * 64 + ni * 32 + l31, qend - 1);
                const int t = div_magic(qc, p.vmagic);
                const int qres = ident ? (t * p.stride + p.res_off) * V + (qc - t * V) : 0;
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const int co = rbase + mi * 32 + 4 * kh + (g & 3) + 8 * (g >> 2);
                    rv[ni][g] = ident ? rseg[(int64_t)min(co, p.Cout - 1) * rcs + qres] : 0.f;
                }
            }
        }
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            float v0 = acc[mi][0][g] + bv[g] + rv[0][g];
            float v1 = acc[mi][1][g] + bv[g] + rv[1][g];
            if (p.relu) { v0 = relu_nan(v0); v1 = relu_nan(v1); }
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v0), __float_as_uint(v1), false, false);
            acc[mi][0][g] = __uint_as_float(sw[0]);        // row rbase + mi*32 + (g&3) + 8(g>>2), this lane's column qb
            acc[mi][1][g] = __uint_as_float(sw[1]);        // row + 4
        }
        if (full) {
            if (qv) {
                const unsigned qo = 4u * (unsigned)qb;
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    float *orow = oseg + (int64_t)(rbase + mi * 32 + (g & 3) + 8 * (g >> 2)) * Q;
                    st_lane(orow, qo, acc[mi][0][g]);
                    st_lane(orow + 4 * (int64_t)Q, qo, acc[mi][1][g]);
                }
            }
        } else {
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int row0 = rbase + mi * 32 + (g & 3) + 8 * (g >> 2);
                if (qv && row0 < p.Cout) oseg[(int64_t)row0 * Q + qb] = acc[mi][0][g];
                if (qv && row0 + 4 < p.Cout) oseg[(int64_t)(row0 + 4) * Q + qb] = acc[mi][1][g];
            }
        }
    }
}
