// gcn_split.hip -- OPT-IN precision mode "bf16x3" of the graph-conv stage (skeleton graphs, C_out a multiple of 128):
//   y = ReLU( W' . agg(x) + b' + gcn_residual(x) )      models/base.py:260-270
// with the channel-mixing GEMM on the bf16 matrix pipe (three bf16 pieces per fp32 operand, six piece products per fp32
// product, fp32 accumulation -- split_core.h); the sparse adjacency aggregation stays exact fp32 on the vector unit and
// its RESULT is what gets split, so the only arithmetic that differs from csk_gcn_stage_f32 is the channel mix.
//
// Same GEMM machinery as tcn_split.hip with the three adjacency subsets in the role of three taps: per 16-channel chunk
//   (1) the raw x rows of the tile's frames go to LDS in quarter-major order  Xs[quarter of 4 channels][position]
//       (16-byte vectors: consecutive lanes = consecutive positions, conflict-free both ways);
//   (2) thread (column, channel half) forms  agg_r = sum_e val_e * x[src_e]  for its 8 channels and the 3 subsets from <= 6
//       register-resident (offset, weight) adjacency entries (<= 1/1/4 non-zeros per column: 12 ds_read_b128, 48 FMAs),
//       splits the 3 x 8 sums into bf16 pieces and writes the operand tiles  Bl[subset][piece][half][column];
//   (3) 3 "taps" x 24 MFMAs per wave against the staged split weights  Wl[subset][piece][half][row].
// 8 waves / one workgroup per CU / 128 x 256 tile as tcn_split.hip; weights single-buffered (the chunk boundary has its
// barriers anyway), 130 KB of LDS.  The conv gcn_residual (C_in != C_out) is a second K phase over x itself (one tap).
#include "split_core.h"

struct GcnSplitParams {
    const float *x, *bias;
    const u32x4 *w, *wres;        // fold.pack_conv_weight_split of (C_out, C_in, 3 subsets) / (C_out, C_in, 1)
    float *y;
    const int32_t *ell_src;
    const float *ell_val;
    int ell_cnt[3], ell_w;
    int Cin, nchunks, Cout, Mpad, frames, V, res_mode, ldx;
    unsigned vmagic, mtiles, qtiles;
    int nt, fast_epi;
};

template <int NS4>
__global__ __launch_bounds__(NTH2, 2) void gcn_split_stage_kernel(const GcnSplitParams p) {
    constexpr int MT = 128, NT = 256, WM = 2;
    constexpr int WSZ = TG * 6 * MT;                        // vectors of the weight buffer (3 subsets)
    extern __shared__ __attribute__((aligned(16))) u32x4 smem4[];
    u32x4 *Wl = smem4;                                      // [3 subsets][3 pieces][2 halves][MT]
    u32x4 *Bl = smem4 + WSZ;                                // [3 subsets][3 pieces][2 halves][NT]
    f32x4 *Xs = reinterpret_cast<f32x4 *>(smem4 + WSZ + TG * 6 * NT);   // [4 channel quarters][ldx positions]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    const int l31 = lane & 31, kh = lane >> 5;
    const unsigned wid = xcd_contiguous_id(blockIdx.x, gridDim.x);
    const int m0 = (int)(wid % p.mtiles) * MT, q0 = (int)((wid / p.mtiles) % p.qtiles) * p.nt;
    const int seg = (int)(wid / (p.mtiles * p.qtiles));
    const int V = p.V, Q = p.frames * V, ldx = p.ldx;
    const int qend = min(q0 + p.nt, Q);
    const int ta = div_magic(q0, p.vmagic), tb = div_magic(qend - 1, p.vmagic);
    const int span = (tb - ta + 1) * V;                     // positions of the tile's (whole) frames

    // aggregation role of this thread: column ac (0..255), channel half ah; <= 6 adjacency entries of its joint
    const int ac = tid & (NT - 1), ah = tid >> 8;
    int eoff[6], ioff;
    float eval[6];
    {
        const int q = min(q0 + ac, qend - 1);
        const int t = div_magic(q, p.vmagic);
        const int w = q - t * V, fb = (t - ta) * V;
        ioff = fb + w;
#pragma unroll
        for (int e = 0; e < 6; ++e) {
            const int r = e < 2 ? e : 2, k = e < 2 ? 0 : e - 2;          // subsets 0,1: one entry; subset 2: up to four
            const bool have = k < p.ell_cnt[r];
            const int idx = (r * V + w) * p.ell_w + min(k, p.ell_w - 1);
            eoff[e] = fb + (have ? p.ell_src[idx] : 0);
            eval[e] = have ? p.ell_val[idx] : 0.f;
        }
    }
    const int off0 = wn * 64 + l31, off1 = off0 + 32;       // this lane's two MFMA columns
    const int offA = wm * 64 + l31;
    const int toff[3] = {0, 6 * NT, 12 * NT};               // operand tile of subset r starts at r * 6 * NT
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[a][b][g] = 0.f;

    WSplitStage<MT> ws;
    // x staging: wave w loads channel half h = w & 1 of the position sweeps (w >> 1) + 4 i; a lane holds the 8 channels of
    // its position (BSplitStage's load pattern, kept raw)
    const int xh = wave & 1;
    unsigned xg[NS4], xl[NS4];
    float xv[NS4][8];
#pragma unroll
    for (int i = 0; i < NS4; ++i) {
        const int j = min(((wave >> 1) + 4 * i) * 64 + lane, span - 1);
        xg[i] = (unsigned)(ta * V + j);
        xl[i] = (unsigned)j;
    }
    const float *seg_base = p.x + (int64_t)seg * p.Cin * Q;
    auto issue_x = [&](int c0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = c0 + 8 * xh + j;
            const float *src = seg_base + (int64_t)min(c, p.Cin - 1) * Q;
            const bool ok = c < p.Cin;
#pragma unroll
            for (int i = 0; i < NS4; ++i) {
                const float v = src[xg[i]];
                xv[i][j] = ok ? v : 0.f;
            }
        }
    };
    auto commit_x = [&]() {
#pragma unroll
        for (int i = 0; i < NS4; ++i) {
            f32x4 lo = {xv[i][0], xv[i][1], xv[i][2], xv[i][3]}, hi = {xv[i][4], xv[i][5], xv[i][6], xv[i][7]};
            Xs[(2 * xh) * ldx + xl[i]] = lo;
            Xs[(2 * xh + 1) * ldx + xl[i]] = hi;
        }
    };
    // aggregated (or, for the residual phase, plain) operand tiles of this thread's (column, half)
    auto put = [&](int r, const float (&s)[8]) {
        bf16x8 ph, pm, pl;
        split8(s, ph, pm, pl);
        u32x4 *dst = Bl + (r * 6 + ah) * NT + ac;
        dst[0] = __builtin_bit_cast(u32x4, ph);
        dst[2 * NT] = __builtin_bit_cast(u32x4, pm);
        dst[4 * NT] = __builtin_bit_cast(u32x4, pl);
    };
    auto gather = [&](int o, float (&d)[8]) {
        const f32x4 a = Xs[(2 * ah) * ldx + o], b = Xs[(2 * ah + 1) * ldx + o];
        d[0] = a[0]; d[1] = a[1]; d[2] = a[2]; d[3] = a[3]; d[4] = b[0]; d[5] = b[1]; d[6] = b[2]; d[7] = b[3];
    };
    auto aggregate = [&]() {
        float x0[8], s[8];
        gather(eoff[0], x0);
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] = eval[0] * x0[j];
        put(0, s);
        gather(eoff[1], x0);
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] = eval[1] * x0[j];
        put(1, s);
        gather(eoff[2], x0);
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] = eval[2] * x0[j];
#pragma unroll
        for (int e = 3; e < 6; ++e) {                      // same order as gcn_stage_sparse2_kernel: fmaf chain over the entries
            gather(eoff[e], x0);
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] = fmaf(eval[e], x0[j], s[j]);
        }
        put(2, s);
    };

    // ---- phase 1: the three adjacency subsets
    {
        const u32x4 *wb = p.w + m0;
        const int64_t sstride = (int64_t)TG * 6 * p.Mpad;
        ws.issue(wb, p.Mpad, tid);
        issue_x(0);
        for (int c = 0; c < p.nchunks; ++c) {
            __syncthreads();                               // MFMAs of chunk c - 1 are done with Wl / Bl, its aggregation with Xs
            ws.commit(Wl, tid);
            commit_x();
            __syncthreads();
            if (c + 1 < p.nchunks) {
                ws.issue(wb + (c + 1) * sstride, p.Mpad, tid);
                issue_x((c + 1) * KS);
            }
            aggregate();
            __syncthreads();
            __builtin_amdgcn_s_setprio(1);
            mfma_split_taps<MT, TG>(Wl, Bl, NT, toff, offA, off0, off1, kh, acc);
            __builtin_amdgcn_s_setprio(0);
        }
    }
    // ---- phase 2: conv gcn_residual (models/base.py:246-254): 1 x 1 conv over x itself, one tap per chunk
    if (p.res_mode == CSK_RES_CONV) {
        const u32x4 *wb = p.wres + m0;
        const int64_t sstride = (int64_t)TG * 6 * p.Mpad;
        ws.issue(wb, p.Mpad, tid);
        issue_x(0);
        for (int c = 0; c < p.nchunks; ++c) {
            __syncthreads();
            ws.commit(Wl, tid);
            commit_x();
            __syncthreads();
            if (c + 1 < p.nchunks) {
                ws.issue(wb + (c + 1) * sstride, p.Mpad, tid);
                issue_x((c + 1) * KS);
            }
            float x0[8];
            gather(ioff, x0);
            put(0, x0);
            __syncthreads();
            mfma_split_taps<MT, 1>(Wl, Bl, NT, toff, offA, off0, off1, kh, acc);
        }
    }
    // ---- epilogue (split_core.h): + bias + identity gcn_residual, ReLU
    SplitEpi e;
    e.bias = p.bias; e.rseg = seg_base; e.oseg = p.y + (int64_t)seg * p.Cout * Q;
    e.Cout = p.Cout; e.Tres = p.frames; e.V = V; e.Q = Q; e.stride = 1; e.res_off = 0; e.relu = 1;
    e.ident = p.res_mode == CSK_RES_IDENTITY; e.fast_epi = p.fast_epi; e.vmagic = p.vmagic;
    split_epilogue<MT>(e, acc, m0, wm, wn, q0, qend, lane);
}

extern "C" int csk_gcn_stage_bf16x3(const float *x, float *y, const void *w_split, const void *w_res_split, const float *bias,
                                    const int32_t *ell_src, const float *ell_val, const int32_t *ell_cnt, int ell_w, int n_seg,
                                    int c_in, int c_out, int frames, int V, int res_mode, void *stream) {
    if (!x || !y || !w_split || !bias || !ell_src || !ell_val || !ell_cnt) CSK_FAIL("gcn_stage_bf16x3: null pointer");
    if (n_seg <= 0 || c_in <= 0 || c_out <= 0 || frames <= 0 || V < 2 || V > 64) CSK_FAIL("gcn_stage_bf16x3: bad dims");
    if (c_out % 128) CSK_FAIL("gcn_stage_bf16x3: built for C_out a multiple of 128 (got %d); use csk_gcn_stage_f32", c_out);
    if (res_mode != CSK_RES_IDENTITY && res_mode != CSK_RES_CONV) CSK_FAIL("gcn_stage_bf16x3: res_mode must be identity or conv");
    if (res_mode == CSK_RES_IDENTITY && c_in != c_out) CSK_FAIL("gcn_stage_bf16x3: identity residual needs c_in == c_out");
    if (res_mode == CSK_RES_CONV && !w_res_split) CSK_FAIL("gcn_stage_bf16x3: conv residual without w_res_split");
    if (ell_w < 1 || ell_w > V || ell_cnt[0] < 0 || ell_cnt[0] > 1 || ell_cnt[1] < 0 || ell_cnt[1] > 1 || ell_cnt[2] < 0 || ell_cnt[2] > 4 ||
        ell_cnt[2] > ell_w)
        CSK_FAIL("gcn_stage_bf16x3: needs a skeleton-sparse adjacency (<= 1/1/4 non-zeros per column)");
    if ((int64_t)frames * V >= (1 << 26)) CSK_FAIL("gcn_stage_bf16x3: frames*V too large for 32-bit position arithmetic");
    if (((uintptr_t)w_split | (uintptr_t)(w_res_split ? w_res_split : w_split)) & 15) CSK_FAIL("gcn_stage_bf16x3: packed weights must be 16-byte aligned");
    GcnSplitParams p;
    p.x = x; p.bias = bias; p.w = (const u32x4 *)w_split; p.wres = (const u32x4 *)w_res_split; p.y = y;
    p.ell_src = ell_src; p.ell_val = ell_val;
    for (int i = 0; i < 3; ++i) p.ell_cnt[i] = ell_cnt[i];
    p.ell_w = ell_w;
    p.Cin = c_in; p.nchunks = round_up(c_in, KS) / KS; p.Cout = c_out; p.Mpad = round_up(c_out, CSK_MT); p.frames = frames; p.V = V;
    p.res_mode = res_mode; p.vmagic = vmagic_of(V);
    p.fast_epi = (int64_t)frames * V < (1ll << 27);
    constexpr int MT = 128, NT = 256;
    p.nt = NT;
    const int max_dt = (NT + V - 2) / V;
    p.ldx = round_up((max_dt + 1) * V, 4);                  // positions of the whole frames a tile touches
    const int nj = (p.ldx + 63) / 64, ns4 = (nj + 3) / 4;
    if (ns4 > 2) CSK_FAIL("gcn_stage_bf16x3: %d joints make the x tile of a 256-position tile longer than 512 positions", V);
    const size_t lds = (size_t)(TG * 6 * MT + TG * 6 * NT + 4 * p.ldx) * 16;
    const int Q = frames * V;
    p.qtiles = (Q + NT - 1) / NT; p.mtiles = p.Mpad / MT;
    if ((int64_t)p.qtiles * p.mtiles * n_seg >= (1ll << 31)) CSK_FAIL("gcn_stage_bf16x3: grid too large");
    void (*kern)(GcnSplitParams) = ns4 <= 1 ? gcn_split_stage_kernel<1> : gcn_split_stage_kernel<2>;
    if (const int e = csk_ensure_lds((const void *)kern, lds)) return e;
    hipLaunchKernelGGL(kern, dim3(p.qtiles * p.mtiles * n_seg), dim3(NTH2), lds, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}
