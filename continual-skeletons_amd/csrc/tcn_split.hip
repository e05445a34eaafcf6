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
// staged 3 taps at a time (36.9 KB per stage, two buffers).
#include "split_core.h"

// Stride: a stride-s temporal conv reads, for tap r, the source frames s t + r - pad: the taps of one residue class
// rho = r mod s read ONE de-interleaved set of frames (s t' + rho - pad), in which consecutive taps are one frame apart --
// a stride-1 conv.  The activation tile therefore holds the classes side by side, each de-interleaved,
//     [ class 0: (tb - ta) + n_0 frames | class 1: (tb - ta) + n_1 frames | ... ]     (n_rho taps in class rho)
// the weights are packed class-major (taps 0, s, 2s, ..., then 1, 1 + s, ...: fold.pack_conv_weight_split), and a tap is
// again a pure address shift -- (frame base of its class + its index in the class) * V -- for every stride: three weight
// stages of three taps per 16-channel chunk whatever the stride, no frame a tap set does not use, no bank conflicts from
// strided frame selection.
struct TcnSplitParams {
    const float *y, *xres, *bias;
    const u32x4 *w, *wres;        // packed split weights (fold.pack_conv_weight_split): [chunk][9 | 3 tap slots][3][2][Mpad] vectors of 8 bf16
    float *out;
    int C, nchunks, Tin, Cout, Mpad, Tout, V, stride, pad;
    int ncls, ntap_cls[4];        // residue classes of the taps and their sizes
    int res_mode, Cres, nchunks_res, Tres, res_off, relu, ldb;
    unsigned vmagic, mtiles, qtiles;
    int nt, fast_epi;
    int diag;   // CSK_DIAG + CSK_SPLIT_SKIP=<bits>: 1 weight staging, 2 activation staging, 4 MFMAs, 8 barriers skipped in the K loop (timing experiments)
};

template <int MT, int NS4>
__global__ __launch_bounds__(NTH2, 2) void tcn_split_stage_kernel(const TcnSplitParams p) {
    constexpr int WM = MT / 64;
    constexpr int WSZ = TG * 6 * MT;                        // vectors of one weight buffer
    constexpr bool PRE = NS4 <= (MT == 128 ? 2 : 3);        // early split of the next tile (where the registers allow it)
    extern __shared__ __attribute__((aligned(16))) u32x4 smem4[];
    u32x4 *Wl0 = smem4;                                     // 2 x [TG][3][2][MT]
    u32x4 *Bl = smem4 + 2 * WSZ;                            // [3][2][ldb]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    const int l31 = lane & 31, kh = lane >> 5;
    const unsigned wid = xcd_contiguous_id(blockIdx.x, gridDim.x);
    const int m0 = (int)(wid % p.mtiles) * MT, q0 = (int)((wid / p.mtiles) % p.qtiles) * p.nt;
    const int seg = (int)(wid / (p.mtiles * p.qtiles));
    const int V = p.V, Q = p.Tout * V;
    const int qend = min(q0 + p.nt, Q);
    const int ta = div_magic(q0, p.vmagic), tb = div_magic(qend - 1, p.vmagic);
    const int dt = tb - ta;

    int off[2];                                             // this lane's two columns inside a class of the tile
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int q = min(q0 + wn * 64 + ni * 32 + l31, qend - 1);
        off[ni] = q - ta * V;
    }
    // LDS offset of tap t (class-major order): (first frame of its class + index in the class) * V -- wave-uniform scalars
    int toff[NSTAGE * TG];
#pragma unroll
    for (int t = 0; t < NSTAGE * TG; ++t) {                 // (static indices only: a runtime-indexed array would live in scratch)
        int rem = t, fb = 0, o = 0;
        bool done = false;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = p.ntap_cls[r];
            if (!done && rem < n) { o = (fb + rem) * V; done = true; }
            if (!done) { rem -= n; fb += dt + n; }
        }
        toff[t] = o;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[a][b][g] = 0.f;
    const int offA = wm * 64 + l31;

    WSplitStage<MT> ws;
    BSplitStage<NS4> bs;
    // ---- phase 1: the 9-tap conv.  Flat sequence of weight stages g = 3 c + s (3 taps each); Wl[g & 1] holds stage g, the
    // registers hold stage g + 1 until it is committed in front of stage g's MFMAs, stage g + 2 is then issued.  The
    // activation tile of chunk c sits in Bl, chunk c + 1 is in flight in registers (split beside the last stage's MFMAs).
    {
        const float *seg_base = p.y + (int64_t)seg * p.C * p.Tin * V;
        const int64_t cs = (int64_t)p.Tin * V;
        const int nchunks = p.nchunks, nst = nchunks * NSTAGE;
        const u32x4 *wb = p.w + m0;
        const int64_t sstride = (int64_t)TG * 6 * p.Mpad;
        bs.setup(p.stride * ta - p.pad, p.stride, dt, p.ncls, p.ntap_cls, p.Tin, V, p.vmagic, lane, wave);
        ws.issue(wb, p.Mpad, tid);
        bs.issue(seg_base, p.C, cs, 0);
        ws.commit(Wl0, tid);
        bs.commit(Bl, p.ldb);
        if (nst > 1) ws.issue(wb + sstride, p.Mpad, tid);
        if (nchunks > 1) bs.issue(seg_base, p.C, cs, KS);
        __syncthreads();
        const bool early = wave < 4 || (p.diag & 16);
        for (int c = 0; c < nchunks; ++c) {
#pragma unroll
            for (int s = 0; s < NSTAGE; ++s) {
                const int g = c * NSTAGE + s;
                u32x4 *cur = Wl0 + (g & 1) * WSZ, *oth = Wl0 + ((g & 1) ^ 1) * WSZ;
                // Stagger: the two waves of a SIMD (w and w + 4) run the stage's vector / memory work and its MFMAs in
                // opposite orders, so one feeds the matrix pipe while the other stages -- both orders are legal inside the
                // barrier interval (the staging writes the OTHER weight buffer and registers only)
                auto stage_work = [&]() {
                    if (g + 1 < nst && !(p.diag & 1)) ws.commit(oth, tid);
                    if (g + 2 < nst && !(p.diag & 1)) ws.issue(wb + (g + 2) * sstride, p.Mpad, tid);
                };
                auto stage_mfma = [&]() {
                    __builtin_amdgcn_s_setprio(1);
                    if (!(p.diag & 4)) mfma_split_taps<MT, TG>(cur, Bl, p.ldb, toff + s * TG, offA, off[0], off[1], kh, acc);
                    __builtin_amdgcn_s_setprio(0);
                };
                if (early) stage_work();
                if (PRE && s == NSTAGE - 1 && c + 1 < nchunks) bs.presplit();      // next tile's pieces, beside this stage's MFMAs
                stage_mfma();
                if (!early) stage_work();
                if (!(p.diag & 8)) __syncthreads();
            }
            if (c + 1 < nchunks && !(p.diag & 2)) {
                if (PRE) bs.commit_pk(Bl, p.ldb);
                else bs.commit(Bl, p.ldb);
                if (c + 2 < nchunks) bs.issue(seg_base, p.C, cs, (c + 2) * KS);
                if (!(p.diag & 8)) __syncthreads();
            }
        }
    }
    // ---- phase 2: 1 x 1 strided residual conv over the block input (models/base.py:372-374): one tap per 16-channel chunk
    if (p.res_mode == CSK_RES_CONV) {
        const float *seg_base = p.xres + (int64_t)seg * p.Cres * p.Tres * V;
        const int64_t cs = (int64_t)p.Tres * V;
        const u32x4 *wb = p.wres + m0;
        const int64_t sstride = (int64_t)TG * 6 * p.Mpad;      // one stage (tap 0 + two zero slots) per chunk
        const int one[4] = {1, 0, 0, 0};
        const int tz[1] = {0};
        bs.setup(p.stride * ta + p.res_off, p.stride, dt, 1, one, p.Tres, V, p.vmagic, lane, wave);
        ws.issue(wb, p.Mpad, tid);
        bs.issue(seg_base, p.Cres, cs, 0);
        for (int c = 0; c < p.nchunks_res; ++c) {
            ws.commit(Wl0, tid);                       // (phase 1 / the previous chunk ended with a barrier)
            bs.commit(Bl, p.ldb);
            __syncthreads();
            if (c + 1 < p.nchunks_res) {
                ws.issue(wb + (c + 1) * sstride, p.Mpad, tid);
                bs.issue(seg_base, p.Cres, cs, (c + 1) * KS);
            }
            mfma_split_taps<MT, 1>(Wl0, Bl, p.ldb, tz, offA, off[0], off[1], kh, acc);
            __syncthreads();
        }
    }
    // ---- epilogue (split_core.h)
    SplitEpi e;
    e.bias = p.bias; e.rseg = p.xres + (int64_t)seg * p.Cres * p.Tres * V; e.oseg = p.out + (int64_t)seg * p.Cout * Q;
    e.Cout = p.Cout; e.Tres = p.Tres; e.V = V; e.Q = Q; e.stride = p.stride; e.res_off = p.res_off; e.relu = p.relu;
    e.ident = p.res_mode == CSK_RES_IDENTITY; e.fast_epi = p.fast_epi; e.vmagic = p.vmagic;
    split_epilogue<MT>(e, acc, m0, wm, wn, q0, qend, lane);
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
    if (stride < 1 || stride > 4 || pad < 0 || pad >= k) CSK_FAIL("tcn_stage_bf16x3: bad stride/pad (stride <= 4)");
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
    p.C = c; p.nchunks = round_up(c, KS) / KS; p.Tin = t_in; p.Cout = c_out; p.Mpad = round_up(c_out, CSK_MT); p.Tout = t_out;
    p.V = V; p.stride = stride; p.pad = pad;
    p.ncls = stride < k ? stride : k;
    for (int r = 0; r < 4; ++r) p.ntap_cls[r] = r < p.ncls ? (k - r + stride - 1) / stride : 0;
    p.res_mode = res_mode; p.Cres = c_res > 0 ? c_res : 1; p.nchunks_res = round_up(p.Cres, KS) / KS;
    p.Tres = t_res > 0 ? t_res : 1; p.res_off = res_off; p.relu = relu;
    p.vmagic = vmagic_of(V);
    p.fast_epi = (int64_t)p.Tres * V < (1ll << 27) && (int64_t)t_out * V < (1ll << 27);
    p.diag = csk_diag_int("CSK_SPLIT_SKIP");
    const bool big = (p.Mpad % 128) == 0;
    const int MT = big ? 128 : 64, NT = 32768 / MT;
    // a tile stages (stride * frames spanned + 9) * V positions per 16-byte row: at most 16 (64-row tiles) / 16 (128-row
    // tiles) x 64 positions (4 sweeps per wave), and the tile must fit the CU's LDS next to the two weight buffers; longer
    // spans (stride 3, many joints) narrow the tile -- never the case for the skeleton shapes
    const size_t wl = (size_t)2 * TG * 6 * MT * 16;
    p.nt = NT;
    for (;;) {
        const int max_dt = (p.nt + V - 2) / V;
        p.ldb = round_up((stride * max_dt + k) * V, 4);
        if (((p.ldb + 63) / 64 <= 16 && wl + (size_t)6 * p.ldb * 16 <= 160 * 1024) || p.nt == 1) break;
        p.nt = p.nt > 16 ? p.nt - 16 : 1;
    }
    const int nj = (p.ldb + 63) / 64;
    const size_t lds = wl + (size_t)6 * p.ldb * 16;
    if (nj > 16 || lds > 160 * 1024) CSK_FAIL("tcn_stage_bf16x3: activation tile of %d positions exceeds the staged maximum", p.ldb);
    const int ns4 = (nj + 3) / 4;
    const int Q = t_out * V;
    p.qtiles = (Q + p.nt - 1) / p.nt; p.mtiles = p.Mpad / MT;
    if ((int64_t)p.qtiles * p.mtiles * n_seg >= (1ll << 31)) CSK_FAIL("tcn_stage_bf16x3: grid too large");
    dim3 grid(p.qtiles * p.mtiles * n_seg);
    void (*kern)(TcnSplitParams);
    if (big) kern = ns4 <= 2 ? tcn_split_stage_kernel<128, 2> : ns4 == 3 ? tcn_split_stage_kernel<128, 3> : tcn_split_stage_kernel<128, 4>;
    else kern = ns4 <= 2 ? tcn_split_stage_kernel<64, 2> : ns4 == 3 ? tcn_split_stage_kernel<64, 3> : tcn_split_stage_kernel<64, 4>;
    if (const int e = csk_ensure_lds((const void *)kern, lds)) return e;
    hipLaunchKernelGGL(kern, grid, dim3(NTH2), lds, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}
