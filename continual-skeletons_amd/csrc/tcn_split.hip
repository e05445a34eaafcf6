// tcn_split.hip -- OPT-IN precision mode "bf16x3" of the temporal-conv stage / step (gfx950 / MI355X).
//
// The default path computes the 9x1 temporal conv in exact fp32 (v_mfma_f32_32x32x2_f32, csrc/tcn.hip, csrc/step.hip):
// that instruction runs at 1/16 of the bf16 MFMA rate, and the exact-fp32 kernels sit within ~10 % of their structural
// ceiling.  This file is the one lever left: fp32-GRADE arithmetic on the bf16 matrix pipe.  Every fp32 operand is split
// into three bf16 pieces  x = h + m + l  (h = bf16(x), m = bf16(x - h), l = bf16(x - h - m): 24 significand bits in all)
// and a product a*b is the sum of the six piece products of order <= 2 (hh, hm, mh, hl, lh, mm; the dropped ones are
// below 2^-24 relative), each one v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  It is NOT fp32 and is never
// reported as such: selected explicitly (blocks.set_precision("bf16x3")), priced under its own bench key with dtype
// "bf16x3-split, f32 accumulate"; the default path and every bitwise test of it are untouched.
//
// GEMM view as tcn.hip:  D[co, q] = sum_r sum_c W[r][c][co] * y[c][q + (r - pad) V]; a K step of the bf16 MFMA is 16
// channels of ONE tap.  Operand images are K-CONTIGUOUS (8 consecutive channels of one row / position = one 16-byte
// lane fragment, read with a single ds_read_b128):
//   Wl [tap in stage][piece][k-half][MT rows][8 ch]     pre-split and laid out like this on the host (fold.py)
//   Bl [piece][k-half][positions][8 ch]                 split at staging time: a lane loads 8 channels of its position
//                                                       (coalesced along positions), splits them and writes 3 x 16 B
// One activation tile (16 channels x span positions) serves all 9 taps (a tap is an address shift of V positions);
// the weights of a 16-channel chunk (9 x 16 x MT x 6 B = 110 KB at MT = 128) do not fit next to it, so they are
// staged in TG taps at a time (TG = 3: 36.9 KB; Bl 36.9 KB -> two workgroups per CU).
#include "mfma_core.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

static constexpr int KS = 16;          // channels per K step / chunk of the split kernels

struct TcnSplitParams {
    const float *y, *xres, *bias;
    const u32x4 *w, *wres;        // packed split weights: [chunk][Kp][3][2][Mpad] vectors of 8 bf16 (fold.pack_conv_weight_split)
    float *out;
    int C, Cpad, Cout, Mpad, Tin, Tout, V, K, Kp, stride, pad;
    int res_mode, Cres, CresPad, Tres, res_off, relu, ldb;
    unsigned vmagic, mtiles, qtiles;
    int nt, fast_epi;
};

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

// weights of one stage (TG taps x 3 pieces x 2 k-halves x MT rows, 16 B each): a contiguous run of rows per
// (tap, piece, half) in global memory, copied as it stands (register prefetch + ds_write_b128)
template <int MT, int TG>
struct WSplitStage {
    static constexpr int NV = TG * 6 * MT;
    static constexpr int WB = (NV + NTHREADS - 1) / NTHREADS;
    unsigned goff[WB], loff[WB];
    u32x4 v[WB];
    __device__ __forceinline__ void setup(int Mpad, int tid) {
#pragma unroll
        for (int u = 0; u < WB; ++u) {
            const int e = min(u * NTHREADS + tid, NV - 1);
            goff[u] = (unsigned)((e / MT) * Mpad + (e % MT));
            loff[u] = (unsigned)e;
        }
    }
    __device__ __forceinline__ void issue(const u32x4 *__restrict__ base) {
#pragma unroll
        for (int u = 0; u < WB; ++u) v[u] = base[goff[u]];
    }
    __device__ __forceinline__ void commit(u32x4 *__restrict__ Wl) const {
#pragma unroll
        for (int u = 0; u < WB; ++u) Wl[loff[u]] = v[u];
    }
};

// activations of one 16-channel chunk: wave w stages k-half h = w & 1 of the position sweeps (w >> 1), (w >> 1) + 2, ...
// (64 positions each): per sweep a lane loads the 8 channels of its position, unconditionally (clamped address + select,
// see BStage in mfma_core.h), and at commit time splits them into the three piece fragments.
template <int NS2>
struct BSplitStage {
    unsigned goff[NS2], loff[NS2], valid;
    int h;
    float v[NS2][8];
    __device__ __forceinline__ void setup(int pbase, int span, int TV, int lane, int wave) {
        h = wave & 1;
        valid = 0;
#pragma unroll
        for (int i = 0; i < NS2; ++i) {
            const int j = min(((wave >> 1) + 2 * i) * 64 + lane, span - 1);
            const int pp = pbase + j;
            goff[i] = (unsigned)min(max(pp, 0), TV - 1);
            loff[i] = (unsigned)j;
            valid |= (pp >= 0 && pp < TV) ? (1u << i) : 0u;
        }
    }
    __device__ __forceinline__ void issue_sweep(int i, const float *__restrict__ seg_base, int C, int64_t cs, int c0) {
        if (i >= NS2) return;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = c0 + 8 * h + j;
            const float x = seg_base[(int64_t)min(c, C - 1) * cs + goff[i]];
            v[i][j] = (c < C && ((valid >> i) & 1u)) ? x : 0.f;
        }
    }
    __device__ __forceinline__ void issue(const float *__restrict__ seg_base, int C, int64_t cs, int c0) {
#pragma unroll
        for (int i = 0; i < NS2; ++i) issue_sweep(i, seg_base, C, cs, c0);
    }
    __device__ __forceinline__ void commit(u32x4 *__restrict__ Bl, int ldb) const {
#pragma unroll
        for (int i = 0; i < NS2; ++i) {
            bf16x8 ph, pm, pl;
            split8(v[i], ph, pm, pl);
            u32x4 *dst = Bl + h * ldb + loff[i];
            dst[0] = __builtin_bit_cast(u32x4, ph);
            dst[2 * ldb] = __builtin_bit_cast(u32x4, pm);
            dst[4 * ldb] = __builtin_bit_cast(u32x4, pl);
        }
    }
};

// nt taps of the staged weights against the activation tile: per tap 12 ds_read_b128 (3 pieces x (2 row + 2 column
// blocks)) and 24 MFMAs (6 piece products x 2 x 2 blocks), small products first
template <int MT>
__device__ __forceinline__ void mfma_split_taps(const u32x4 *__restrict__ Wl, const u32x4 *__restrict__ Bl, int nt, int ldb,
                                                int tapB, int offA, int off0, int off1, int kh, f32x16 (&acc)[2][2]) {
    for (int t = 0; t < nt; ++t) {
        const u32x4 *wr = Wl + (t * 6 + kh) * MT + offA;
        const u32x4 *br = Bl + kh * ldb + t * tapB;
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

template <int MT, int NS2, int TG>
__global__ __launch_bounds__(NTHREADS, 2) void tcn_split_stage_kernel(const TcnSplitParams p) {
    constexpr int NT = 16384 / MT;
    constexpr int WM = MT / 64;
    constexpr int S = (9 + TG - 1) / TG;                    // weight stages of the 9-tap phase
    extern __shared__ __attribute__((aligned(16))) u32x4 smem4[];
    u32x4 *Wl = smem4;                                      // [TG][3][2][MT]
    u32x4 *Bl = smem4 + TG * 6 * MT;                        // [3][2][ldb]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    const int l31 = lane & 31, kh = lane >> 5;
    const unsigned wid = xcd_contiguous_id(blockIdx.x, gridDim.x);
    const int m0 = (int)(wid % p.mtiles) * MT, q0 = (int)((wid / p.mtiles) % p.qtiles) * p.nt;
    const int seg = (int)(wid / (p.mtiles * p.qtiles));
    const int V = p.V, Q = p.Tout * V;
    const int qend = min(q0 + p.nt, Q);
    const int ta = div_magic(q0, p.vmagic), tb = div_magic(qend - 1, p.vmagic);

    int off[2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int q = min(q0 + wn * 64 + ni * 32 + l31, qend - 1);
        const int t = div_magic(q, p.vmagic);
        off[ni] = p.stride * (t - ta) * V + (q - t * V);
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[a][b][g] = 0.f;
    const int offA = wm * 64 + l31;

    WSplitStage<MT, TG> ws;
    BSplitStage<NS2> bs;
    ws.setup(p.Mpad, tid);
    // ---- phase 1: 9 x 1 temporal conv over y, chunks of 16 channels, S weight stages per chunk
    {
        const int fa = p.stride * ta - p.pad;
        const int span = (p.stride * (tb - ta) + p.K) * V;
        const float *seg_base = p.y + (int64_t)seg * p.C * p.Tin * V;
        const int64_t cs = (int64_t)p.Tin * V;
        const u32x4 *wbase = p.w + m0;
        const int64_t stage_stride = (int64_t)TG * 6 * p.Mpad, chunk_stride = (int64_t)p.Kp * 6 * p.Mpad;
        bs.setup(fa * V, span, p.Tin * V, lane, wave);
        ws.issue(wbase);
        bs.issue(seg_base, p.C, cs, 0);
        const int nchunks = p.Cpad / KS;
        for (int c = 0; c < nchunks; ++c) {
#pragma unroll
            for (int s = 0; s < S; ++s) {
                __syncthreads();                       // the previous stage's LDS reads are done
                ws.commit(Wl);
                if (s == 0) bs.commit(Bl, p.ldb);
                __syncthreads();
                // next stage's weights; the next chunk's activations go out one sweep per stage
                const bool last = s == S - 1;
                const int cn = last ? c + 1 : c, sn = last ? 0 : s + 1;
                if (cn < nchunks) ws.issue(wbase + cn * chunk_stride + sn * stage_stride);
                if (c + 1 < nchunks) {
#pragma unroll
                    for (int i = s; i < NS2; i += S) bs.issue_sweep(i, seg_base, p.C, cs, (c + 1) * KS);
                }
                const int ntap = min(TG, p.K - s * TG);
                __builtin_amdgcn_s_setprio(1);
                mfma_split_taps<MT>(Wl, Bl + s * TG * V, ntap, p.ldb, V, offA, off[0], off[1], kh, acc);
                __builtin_amdgcn_s_setprio(0);
            }
        }
    }
    // ---- phase 2: 1 x 1 strided residual conv over the block input (models/base.py:372-374): one tap, one stage per chunk
    if (p.res_mode == CSK_RES_CONV) {
        const int fa = p.stride * ta + p.res_off;
        const int span = (p.stride * (tb - ta) + 1) * V;
        const float *seg_base = p.xres + (int64_t)seg * p.Cres * p.Tres * V;
        const int64_t cs = (int64_t)p.Tres * V;
        const u32x4 *wbase = p.wres + m0;
        const int64_t chunk_stride = (int64_t)6 * p.Mpad;      // Kp = 1
        bs.setup(fa * V, span, p.Tres * V, lane, wave);
        ws.issue(wbase);
        bs.issue(seg_base, p.Cres, cs, 0);
        const int nchunks = p.CresPad / KS;
        for (int c = 0; c < nchunks; ++c) {
            __syncthreads();
            ws.commit(Wl);                              // (surplus vectors of the stage are clamped duplicates / pad taps: never read)
            bs.commit(Bl, p.ldb);
            __syncthreads();
            if (c + 1 < nchunks) {
                ws.issue(wbase + (c + 1) * chunk_stride);
                bs.issue(seg_base, p.Cres, cs, (c + 1) * KS);
            }
            mfma_split_taps<MT>(Wl, Bl, 1, p.ldb, V, offA, off[0], off[1], kh, acc);
        }
    }
    // ---- epilogue: + bias (+ identity residual), ReLU, stores -- the scheme of tcn_stage_kernel (scalar row bases + 32-bit
    // lane offsets on full tiles, permlane32_swap for 256-B row segments), operands loaded one 32-row half at a time
    float *oseg = p.out + (int64_t)seg * p.Cout * Q;
    const float *rseg = p.xres + (int64_t)seg * p.Cres * p.Tres * V;
    const int64_t rcs = (int64_t)p.Tres * V;
    const bool ident = p.res_mode == CSK_RES_IDENTITY;
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

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" int csk_tcn_stage_bf16x3(const float *y, const void *w_split, const float *x_res, const void *w_res_split,
                                    const float *bias, float *out, int n_seg, int c, int c_out, int t_in, int V, int k,
                                    int stride, int pad, int res_mode, int c_res, int t_res, int res_off, int relu,
                                    void *stream) {
    if (!y || !w_split || !bias || !out) CSK_FAIL("tcn_stage_bf16x3: null pointer");
    if (n_seg <= 0 || c <= 0 || c_out <= 0 || t_in <= 0 || V < 2 || V > 64) CSK_FAIL("tcn_stage_bf16x3: bad dims");
    if (k != 9) CSK_FAIL("tcn_stage_bf16x3: the split kernel is built for the 9 x 1 temporal conv (k = %d); use csk_tcn_stage_f32", k);
    if (stride < 1 || pad < 0 || pad >= k) CSK_FAIL("tcn_stage_bf16x3: bad stride/pad");
    if (t_in + 2 * pad < k) CSK_FAIL("tcn_stage_bf16x3: t_in too short for kernel");
    const int t_out = (t_in + 2 * pad - k) / stride + 1;
    if (res_mode != CSK_RES_NONE) {
        if (!x_res) CSK_FAIL("tcn_stage_bf16x3: residual requested without x_res");
        if (res_mode == CSK_RES_IDENTITY && c_res != c_out) CSK_FAIL("tcn_stage_bf16x3: identity residual needs c_res == c_out");
        if (res_mode == CSK_RES_CONV && !w_res_split) CSK_FAIL("tcn_stage_bf16x3: conv residual without w_res");
        if ((t_out - 1) * stride + res_off >= t_res || res_off < 0) CSK_FAIL("tcn_stage_bf16x3: residual frames out of range");
    }
    if ((int64_t)t_in * V >= (1 << 26)) CSK_FAIL("tcn_stage_bf16x3: T*V too large for 32-bit position arithmetic");
    if (((uintptr_t)w_split | (uintptr_t)(w_res_split ? w_res_split : w_split)) & 15) CSK_FAIL("tcn_stage_bf16x3: packed weights must be 16-byte aligned");
    TcnSplitParams p;
    p.y = y; p.w = (const u32x4 *)w_split; p.xres = x_res ? x_res : y; p.wres = (const u32x4 *)w_res_split; p.bias = bias; p.out = out;
    p.C = c; p.Cpad = round_up(c, KS); p.Cout = c_out; p.Mpad = round_up(c_out, CSK_MT);
    p.Tin = t_in; p.Tout = t_out; p.V = V; p.K = k; p.Kp = CSK_SPLIT_KP; p.stride = stride; p.pad = pad;
    p.res_mode = res_mode; p.Cres = c_res > 0 ? c_res : 1; p.CresPad = round_up(p.Cres, KS);
    p.Tres = t_res > 0 ? t_res : 1; p.res_off = res_off; p.relu = relu; p.vmagic = vmagic_of(V);
    p.fast_epi = (int64_t)p.Tres * V < (1ll << 27) && (int64_t)t_out * V < (1ll << 27);
    const bool big = (p.Mpad % 128) == 0;
    const int MT = big ? 128 : 64, NT = 16384 / MT;
    // activation sweeps: at most 10 x 64 positions of a row are staged (5 per wave: the widest spill-free instantiation);
    // longer input spans (stride 3 with many joints, ...) narrow the tile -- never the case for the ST-GCN shapes
    p.nt = NT;
    for (;;) {
        const int max_dt = (p.nt + V - 2) / V;
        p.ldb = round_up((stride * max_dt + k) * V, 4);
        if ((p.ldb + 63) / 64 <= 10 || p.nt == 1) break;
        p.nt = p.nt > 16 ? p.nt - 16 : 1;
    }
    const int nj = (p.ldb + 63) / 64;
    if (nj > 10) CSK_FAIL("tcn_stage_bf16x3: activation tile of %d positions exceeds the staged maximum (640)", p.ldb);
    const int ns2 = (nj + 1) / 2;
    // weight stage depth: 3 taps if the tile then still fits two workgroups per CU (80 KB each), else 2 (also for the
    // 128-row tiles with more than 3 sweeps per wave: their 3-tap form spills)
    const size_t bl = (size_t)6 * p.ldb * 16;
    const int TG = ((size_t)3 * 6 * MT * 16 + bl <= 80 * 1024 && !(big && ns2 > 3)) ? 3 : 2;
    const size_t lds = (size_t)TG * 6 * MT * 16 + bl;
    if (lds > 160 * 1024) CSK_FAIL("tcn_stage_bf16x3: LDS tile %zu B exceeds 160 KiB", lds);
    const int Q = t_out * V;
    p.qtiles = (Q + p.nt - 1) / p.nt; p.mtiles = p.Mpad / MT;
    if ((int64_t)p.qtiles * p.mtiles * n_seg >= (1ll << 31)) CSK_FAIL("tcn_stage_bf16x3: grid too large");
    dim3 grid(p.qtiles * p.mtiles * n_seg);
    void (*kern)(TcnSplitParams);
    if (big) kern = TG == 3 ? tcn_split_stage_kernel<128, 3, 3> : (ns2 <= 3 ? tcn_split_stage_kernel<128, 3, 2> : tcn_split_stage_kernel<128, 5, 2>);
    else kern = TG == 3 ? (ns2 <= 3 ? tcn_split_stage_kernel<64, 3, 3> : tcn_split_stage_kernel<64, 5, 3>)
                        : (ns2 <= 3 ? tcn_split_stage_kernel<64, 3, 2> : tcn_split_stage_kernel<64, 5, 2>);
    if (const int e = csk_ensure_lds((const void *)kern, lds)) return e;
    hipLaunchKernelGGL(kern, grid, dim3(NTHREADS), lds, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}
