// stages.hip -- gfx950 (MI355X) kernels for the ST-GCN clip forward path.
//
// Two stage kernels per SpatioTemporalBlock, both built on one fp32-MFMA "shifted GEMM" core:
//
//   gcn_stage : y   = ReLU( W' . agg(x) + b' + gcn_residual(x) )          models/base.py:260-270
//   tcn_stage : out = ReLU( W' . taps(y) + b' + block_residual(x) )        models/base.py:302-304,376-387
//
// GEMM view (per skeleton sequence = "segment"):  D[co, q] = sum_r sum_c W[r][c][co] * B_r[c][q]
//   q  = flattened (frame, joint) position, V innermost  -> coalesced HBM rows, conflict-free LDS reads
//   TCN: B_r[c][q] = y[c][q + (r - pad) * V]      -- a tap is an address shift inside one LDS tile
//   GCN: B_r[c][q] = sum_v x[c][frame(q), v] * A_eff[r][v, joint(q)]   -- sparse (ELL) VALU gather
//        into an LDS tile, adjacency tables staged in LDS
// Arithmetic: exact fp32 (v_mfma_f32_32x32x2_f32), BatchNorm(eval)/biases folded into W'/b' on the host.
//
// Tiling: 256 threads = 4 waves, each wave owns a 64x64 output tile (2x2 MFMA 32x32 accumulators);
// workgroup tile MT x NT with MT*NT = 16384 (64x256 for C_out = 64, 128x128 otherwise).  K loop walks
// channel chunks of KC = 8; one activation chunk in LDS serves all 9 taps.
#include <stdlib.h>
#include <string.h>

#include "mfma_core.h"

#include <mutex>
#include <unordered_map>

int csk_ensure_lds(const void *kernel, size_t bytes) {
    static std::mutex mu;
    static std::unordered_map<const void *, size_t> cap;
    std::lock_guard<std::mutex> lock(mu);
    auto it = cap.find(kernel);
    if (it != cap.end() && it->second >= bytes) return 0;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) cap[kernel] = bytes;
    return (int)e;
}

static thread_local char g_err[256] = "";
char *csk_err_buf() { return g_err; }
extern "C" int csk_abi_version(void) { return CSK_ABI_VERSION; }
extern "C" const char *csk_last_error(void) { return g_err; }

// ------------------------------------------------------------------------------------------------
// TCN stage
// ------------------------------------------------------------------------------------------------
struct TcnParams {
    const float *y, *w, *xres, *wres, *bias;
    float *out;
    int C, Cpad, Cout, Mpad, Tin, Tout, V, K, stride, pad;
    int res_mode, Cres, CresPad, Tres, res_off, relu, ldb;
    unsigned vmagic, mtiles, qtiles;
    int prio;                     // raise wave priority inside MFMA segments (diagnostic CSK_NOPRIO=1 turns it off)
    unsigned long long *stamps;   // diagnostic (env CSK_STAMPS=<device ptr>): s_memtime stamps per workgroup, see tools/stamp_probe.py
};

template <int MT, int NJ>
__global__ __launch_bounds__(NTHREADS, 2) void tcn_stage_kernel(const TcnParams p) {
    constexpr int NT = 16384 / MT;
    constexpr int WM = MT / 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Wl = smem;
    float *Bl = smem + p.K * KC * MT;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    const int l31 = lane & 31, kh = lane >> 5;
    // work item -> (m-tile fastest: shares the activation tile; then position tile: shares halos; then segment)
    const unsigned wid = xcd_contiguous_id(blockIdx.x, gridDim.x);
    const int m0 = (int)(wid % p.mtiles) * MT, q0 = (int)((wid / p.mtiles) % p.qtiles) * NT;
    const int seg = (int)(wid / (p.mtiles * p.qtiles));
    const int V = p.V, Q = p.Tout * V;
    const int qend = min(q0 + NT, Q);
    const int ta = div_magic(q0, p.vmagic), tb = div_magic(qend - 1, p.vmagic);

    int off[2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int q = min(q0 + wn * 64 + ni * 32 + l31, Q - 1);
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
    unsigned long long st0 = 0, st1 = 0, st2 = 0;
    if (p.stamps) st0 = __builtin_amdgcn_s_memtime();
    WStage<MT> ws;
    BStage<NJ> bs;
    // epilogue operands: 32 biases + (identity residual) 64 block-input values per lane.  They are loaded
    // UNDER THE LAST CHUNK'S MFMAs (the K loops are peeled by one iteration; the staging registers are dead
    // there), unconditionally (clamped indices, bias padded to Mpad), so the epilogue itself is stores only.
    float *oseg = p.out + (int64_t)seg * p.Cout * Q;
    const float *rseg = p.xres + (int64_t)seg * p.Cres * p.Tres * V;
    const int64_t rcs = (int64_t)p.Tres * V;
    const bool ident = p.res_mode == CSK_RES_IDENTITY;
    float bv[2][16], rv[2][2][16];
    auto issue_epilogue_loads = [&]() {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int g = 0; g < 16; ++g) bv[mi][g] = p.bias[m0 + wm * 64 + mi * 32 + 4 * kh + (g & 3) + 8 * (g >> 2)];
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int qc = min(q0 + wn * 64 + ni * 32 + l31, Q - 1);
            const int t = div_magic(qc, p.vmagic);
            const int qres = ident ? (t * p.stride + p.res_off) * V + (qc - t * V) : 0;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const int co = m0 + wm * 64 + mi * 32 + 4 * kh + (g & 3) + 8 * (g >> 2);
                    rv[ni][mi][g] = ident ? rseg[(int64_t)min(co, p.Cout - 1) * rcs + qres] : 0.f;
                }
        }
    };
    const bool conv_res = p.res_mode == CSK_RES_CONV;
    // ---- phase 1: k x 1 temporal conv over y
    {
        const int fa = p.stride * ta - p.pad;
        const int span = (p.stride * (tb - ta) + p.K) * V;
        const float *seg_base = p.y + (int64_t)seg * p.C * p.Tin * V;
        const int64_t cs = (int64_t)p.Tin * V;
        const float *wbase = p.w + m0;
        ws.setup(p.K, p.Cpad, p.Mpad, tid);
        bs.setup(fa * V, span, p.Tin * V, lane);
        ws.issue(wbase);
        bs.issue(seg_base, p.C, cs, 0, wave);
        int c0 = 0;
        unsigned long long ph0 = 0, ph1 = 0, ph2 = 0, ph3 = 0, tq = 0;   // diagnostic phase sums (p.stamps only)
        for (; c0 + KC < p.Cpad; c0 += KC) {
            if (p.stamps) tq = __builtin_amdgcn_s_memtime();
            __syncthreads();                       // previous chunk's LDS reads are done
            if (p.stamps) { const unsigned long long t = __builtin_amdgcn_s_memtime(); ph0 += t - tq; tq = t; }
            ws.commit(Wl);
            bs.commit(Bl, p.ldb, wave);
            __syncthreads();
            if (p.stamps) { const unsigned long long t = __builtin_amdgcn_s_memtime(); ph1 += t - tq; tq = t; }
            if (p.stamps && c0 == 0) st1 = __builtin_amdgcn_s_memtime();
            {
                // the next chunk's loads are issued in three bursts of 9 between three tap segments (see mfma_taps)
                const float *wnext = wbase + (size_t)(c0 + KC) * p.Mpad;
                const int cn = c0 + KC, t1 = (p.K + 2) / 3, t2 = min(p.K, 2 * t1);
                if (p.stamps) { const unsigned long long t = __builtin_amdgcn_s_memtime(); ph2 += t - tq; tq = t; }
                issue_third<0>(ws, bs, wnext, seg_base, p.C, cs, cn, wave);
                // raised priority while in an MFMA segment: this wave then wins issue arbitration against the
                // SIMD partner's commit / load-issue phase (+2 % measured)
                if (p.prio) __builtin_amdgcn_s_setprio(1);
                mfma_taps<MT>(Wl, Bl, 0, t1, p.ldb, V, offA, off[0], off[1], kh, acc);
                __builtin_amdgcn_s_setprio(0);
                issue_third<1>(ws, bs, wnext, seg_base, p.C, cs, cn, wave);
                if (p.prio) __builtin_amdgcn_s_setprio(1);
                if (t1 < t2) mfma_taps<MT>(Wl, Bl, t1, t2, p.ldb, V, offA, off[0], off[1], kh, acc);
                __builtin_amdgcn_s_setprio(0);
                issue_third<2>(ws, bs, wnext, seg_base, p.C, cs, cn, wave);
                if (p.prio) __builtin_amdgcn_s_setprio(1);
                if (t2 < p.K) mfma_taps<MT>(Wl, Bl, t2, p.K, p.ldb, V, offA, off[0], off[1], kh, acc);
                __builtin_amdgcn_s_setprio(0);
            }
            if (p.stamps) { const unsigned long long t = __builtin_amdgcn_s_memtime(); ph3 += t - tq; tq = t; }
        }
        if (p.stamps && lane == 0) {
            unsigned long long *o = p.stamps + (size_t)gridDim.x * 6 + ((size_t)blockIdx.x * 4 + wave) * 4;
            o[0] = ph0; o[1] = ph1; o[2] = ph2; o[3] = ph3;
        }
        __syncthreads();                           // peeled last chunk
        ws.commit(Wl);
        bs.commit(Bl, p.ldb, wave);
        __syncthreads();
        if (!conv_res) issue_epilogue_loads();
        mfma_chunk<MT>(Wl, Bl, p.K, p.ldb, V, offA, off[0], off[1], kh, acc);
    }
    // ---- phase 2: 1x1 strided residual conv over the block input (models/base.py:372-374)
    if (conv_res) {
        const int fa = p.stride * ta + p.res_off;
        const int span = (p.stride * (tb - ta) + 1) * V;
        const float *seg_base = p.xres + (int64_t)seg * p.Cres * p.Tres * V;
        const int64_t cs = (int64_t)p.Tres * V;
        const float *wbase = p.wres + m0;
        ws.setup(1, p.CresPad, p.Mpad, tid);
        bs.setup(fa * V, span, p.Tres * V, lane);
        ws.issue(wbase);
        bs.issue(seg_base, p.Cres, cs, 0, wave);
        int c0 = 0;
        for (; c0 + KC < p.CresPad; c0 += KC) {
            __syncthreads();
            ws.commit(Wl);
            bs.commit(Bl, p.ldb, wave);
            __syncthreads();
            ws.issue(wbase + (size_t)(c0 + KC) * p.Mpad);
            bs.issue(seg_base, p.Cres, cs, c0 + KC, wave);
            mfma_chunk<MT>(Wl, Bl, 1, p.ldb, V, offA, off[0], off[1], kh, acc);
        }
        __syncthreads();
        ws.commit(Wl);
        bs.commit(Bl, p.ldb, wave);
        __syncthreads();
        issue_epilogue_loads();
        mfma_chunk<MT>(Wl, Bl, 1, p.ldb, V, offA, off[0], off[1], kh, acc);
    }
    if (p.stamps) st2 = __builtin_amdgcn_s_memtime();
    // ---- epilogue: + bias (+ identity residual), ReLU, predicated stores.
    // C/D map: col = lane&31, row = (g&3) + 8(g>>2) + 4(lane>>5).  A plain store of accumulator register g
    // writes two 128-B half rows (rows r and r+4).  v_permlane32_swap of the ni=0 / ni=1 registers gives each
    // lane half the SAME row instead: lanes 0-31 columns 0-31, lanes 32-63 columns 32-63 of row r (first
    // result) and of row r+4 (second) -> every store instruction writes one 256-B contiguous row segment.
    {
        const int qb = q0 + wn * 64 + lane;                    // column of this lane after the swap
        const bool qv = qb < Q;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int co_own = m0 + wm * 64 + mi * 32 + 4 * kh + (g & 3) + 8 * (g >> 2);
                float v0 = acc[mi][0][g] + bv[mi][g] + rv[0][mi][g];
                float v1 = acc[mi][1][g] + bv[mi][g] + rv[1][mi][g];
                if (p.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v0), __float_as_uint(v1), false, false);
                // sw[0]: row (co_own - 4*kh), this lane's column qb; sw[1]: row (co_own - 4*kh + 4)
                const int row0 = co_own - 4 * kh;
                if (qv && row0 < p.Cout) oseg[(int64_t)row0 * Q + qb] = __uint_as_float(sw[0]);
                if (qv && row0 + 4 < p.Cout) oseg[(int64_t)(row0 + 4) * Q + qb] = __uint_as_float(sw[1]);
            }
    }
    if (p.stamps && tid == 0) {
        unsigned long long st3 = __builtin_amdgcn_s_memtime();
        unsigned long long *o = p.stamps + (size_t)blockIdx.x * 6;
        o[0] = st0; o[1] = st1; o[2] = st2; o[3] = st3;
        o[4] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));    // HW_REG_HW_ID
        o[5] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));   // HW_REG_XCC_ID
    }
}

// ------------------------------------------------------------------------------------------------
// GCN stage
// ------------------------------------------------------------------------------------------------
struct GcnParams {
    const float *x, *w, *bias;
    float *y;
    const int32_t *ell_src;
    const float *ell_val;
    int ell_cnt[3];
    int ell_w;
    int64_t adj_seg_stride, x_seg_stride, x_chan_stride, y_seg_stride, y_chan_stride;
    int Cin, CinPad, Cout, Mpad, frames, V, R, res_mode, ldb;
    unsigned vmagic, mtiles, qtiles;
    int dense;   // src[e] == e for all subsets and columns (checked on the host side of the ABI by construction)
    int adj_per_frame;   // the (dense) adjacency varies per FRAME of a segment: index = seg * frames + frame
    int lds_frames;      // frames of adjacency staged per workgroup in that mode
};

template <int MT, int NJ>
__global__ __launch_bounds__(NTHREADS, 2) void gcn_stage_kernel(const GcnParams p) {
    constexpr int NT = 16384 / MT;
    constexpr int WM = MT / 64;
    constexpr int TPC = NTHREADS / NT;   // threads per column in the aggregation pass (1 or 2)
    constexpr int KPT = KC / TPC;        // channels per thread there
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int V = p.V, R = p.R, EW = p.ell_w;
    float *Wl = smem;                          // [R][KC][MT]
    float *Xa = Wl + R * KC * MT;              // [R][KC][NT]   aggregated operand
    float *Bx = Xa + R * KC * NT;              // [KC][ldb]     raw x frames
    float *Lv = Bx + KC * p.ldb;               // [3][V][EW]    adjacency values
    int *Ls = reinterpret_cast<int *>(Lv + 3 * V * EW * (p.adj_per_frame ? p.lds_frames : 1));   // [3][V][EW] row indices

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    const int l31 = lane & 31, kh = lane >> 5;
    const unsigned wid = xcd_contiguous_id(blockIdx.x, gridDim.x);
    const int m0 = (int)(wid % p.mtiles) * MT, q0 = (int)((wid / p.mtiles) % p.qtiles) * NT;
    const int seg = (int)(wid / (p.mtiles * p.qtiles));
    const int Q = p.frames * V;
    const int qend = min(q0 + NT, Q);
    const int ta = div_magic(q0, p.vmagic), tb = div_magic(qend - 1, p.vmagic);
    const int span = (tb - ta + 1) * V;

    // adjacency -> LDS, once per workgroup: the fixed graph, this segment's attention matrices, or (per-frame
    // mode, continual A-GCN where every frame is another skeleton) the matrices of the frames this tile touches
    const int adj_n = 3 * V * EW;
    {
        const int nmat = p.adj_per_frame ? (tb - ta + 1) : 1;
        const int64_t first = p.adj_per_frame ? ((int64_t)seg * p.frames + ta) : (int64_t)seg;
        const float *gv = p.ell_val + first * p.adj_seg_stride;
        const int32_t *gs = p.ell_src;             // the index pattern is shared; only the values vary
        for (int e = tid; e < adj_n * nmat; e += NTHREADS) Lv[e] = gv[e];
        for (int e = tid; e < adj_n; e += NTHREADS) Ls[e] = gs[e];
    }
    // dense mode: every subset lists all V source joints in order (src[e] == e), as the A-GCN host code builds it
    const bool dense_all = p.dense;
    // aggregation-pass coordinates of this thread: one column, KPT channels
    const int aj = tid % NT, ak0 = (tid / NT) * KPT;
    const int aq = min(q0 + aj, Q - 1);
    const int at = div_magic(aq, p.vmagic);
    const int aw = aq - at * V;
    const int afb = (at - ta) * V;

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[a][b][g] = 0.f;

    const int offA = wm * 64 + l31;
    const int off0 = wn * 64 + l31, off1 = off0 + 32;
    const float *seg_base = p.x + (int64_t)seg * p.x_seg_stride;

    WStage<MT> ws;
    BStage<NJ> bs;
    const float *wbase = p.w + m0;
    ws.setup(R, p.CinPad, p.Mpad, tid);
    bs.setup(ta * V, span, Q, lane);
    ws.issue(wbase);
    bs.issue(seg_base, p.Cin, p.x_chan_stride, 0, wave);
    for (int c0 = 0; c0 < p.CinPad; c0 += KC) {
        __syncthreads();                           // previous chunk's MFMA reads of Wl / Xa are done
        ws.commit(Wl);
        bs.commit(Bx, p.ldb, wave);
        __syncthreads();
        if (c0 + KC < p.CinPad) {                  // prefetch the next chunk underneath aggregation + MFMA
            ws.issue(wbase + (size_t)(c0 + KC) * p.Mpad);
            bs.issue(seg_base, p.Cin, p.x_chan_stride, c0 + KC, wave);
        }
        // adjacency aggregation: Xa[r][kk][j] = sum_e val_r[e] * Bx[kk][frame(j) + src_r[e]]
        const float *bx = Bx + ak0 * p.ldb + afb;
        if (dense_all) {
            // dense adjacency (A-GCN): all three subsets share the index pattern src = e, so every x value is
            // loaded ONCE and feeds the three subsets' accumulators (11 LDS reads per 24 FMAs instead of 30)
            float s0[KPT], s1[KPT], s2[KPT];
#pragma unroll
            for (int kk = 0; kk < KPT; ++kk) s0[kk] = s1[kk] = s2[kk] = 0.f;
            const int fo = p.adj_per_frame ? (at - ta) * adj_n : 0;          // this column's frame matrix
            const int eb0 = fo + aw * EW, eb1 = fo + (V + aw) * EW, eb2 = fo + (2 * V + aw) * EW;
            for (int e = 0; e < V; ++e) {
                const float v0 = Lv[eb0 + e], v1 = Lv[eb1 + e], v2 = Lv[eb2 + e];
#pragma unroll
                for (int kk = 0; kk < KPT; ++kk) {
                    const float xv = bx[kk * p.ldb + e];
                    s0[kk] = fmaf(v0, xv, s0[kk]);
                    s1[kk] = fmaf(v1, xv, s1[kk]);
                    s2[kk] = fmaf(v2, xv, s2[kk]);
                }
            }
#pragma unroll
            for (int kk = 0; kk < KPT; ++kk) {
                Xa[(0 * KC + ak0 + kk) * NT + aj] = s0[kk];
                Xa[(1 * KC + ak0 + kk) * NT + aj] = s1[kk];
                Xa[(2 * KC + ak0 + kk) * NT + aj] = s2[kk];
            }
        } else {
            for (int r = 0; r < 3; ++r) {
                float s[KPT];
#pragma unroll
                for (int kk = 0; kk < KPT; ++kk) s[kk] = 0.f;
                const int cnt = p.ell_cnt[r];
                const int eb = (r * V + aw) * EW;
                for (int e = 0; e < cnt; ++e) {
                    const int src = Ls[eb + e];
                    const float val = Lv[eb + e];
#pragma unroll
                    for (int kk = 0; kk < KPT; ++kk) s[kk] = fmaf(val, bx[kk * p.ldb + src], s[kk]);
                }
#pragma unroll
                for (int kk = 0; kk < KPT; ++kk) Xa[(r * KC + ak0 + kk) * NT + aj] = s[kk];
            }
        }
        if (R == 4) {   // conv gcn_residual rides the same GEMM as a 4th "subset" with identity adjacency
#pragma unroll
            for (int kk = 0; kk < KPT; ++kk) Xa[(3 * KC + ak0 + kk) * NT + aj] = bx[kk * p.ldb + aw];
        }
        __syncthreads();
        mfma_chunk<MT>(Wl, Xa, R, NT, KC * NT, offA, off0, off1, kh, acc);
    }

    float *oseg = p.y + (int64_t)seg * p.y_seg_stride;
    const bool ident = p.res_mode == CSK_RES_IDENTITY;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int q = q0 + wn * 64 + ni * 32 + l31;
        const bool qv = q < Q;
        const int qc = min(q, Q - 1);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const int cb = m0 + wm * 64 + mi * 32 + 4 * kh;
            float bv[16], rv[16];
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int co = cb + (g & 3) + 8 * (g >> 2);
                bv[g] = p.bias[co];
                rv[g] = ident ? seg_base[(int64_t)min(co, p.Cout - 1) * p.x_chan_stride + qc] : 0.f;
            }
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int co = cb + (g & 3) + 8 * (g >> 2);
                const float v = fmaxf(acc[mi][ni][g] + bv[g] + rv[g], 0.f);
                if (qv && co < p.Cout) oseg[(int64_t)co * p.y_chan_stride + q] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// GCN stage, sparse-graph fast path: the aggregated B operand is formed ON THE FLY in the MFMA loop.
// For skeleton graphs A_eff has <= 1 / 1 / 4 non-zeros per column in the self / inward / outward subsets
// (NTU-25: 1/1/4, OpenPose-18: 1/1/3), so every lane keeps the <= 6 (LDS offset, weight) pairs of its two
// output columns in registers and builds   B_r[c][q] = sum_e val * x[c][frame(q), src_e]   with <= 6 LDS reads
// + FMAs per k-step, against 12-16 MFMAs (768-1024 cycles) that consume them.  No aggregated tile, no
// aggregation phase, one barrier pair per 16 channels.  Dense / per-sample adjacencies (A-GCN) use the
// general kernel above.
// ------------------------------------------------------------------------------------------------
static constexpr int KCG = CSK_CPAD;    // channels per barrier pair == the zero-padding granule of the packed weights

template <int MT, bool CONVRES>
__global__ __launch_bounds__(NTHREADS, 2) void gcn_stage_sparse_kernel(const GcnParams p) {
    constexpr int NT = 16384 / MT;
    constexpr int WM = MT / 64;
    constexpr int R = CONVRES ? 4 : 3;
    constexpr int M4 = MT / 4;
    constexpr int WB = R * KCG * M4 / NTHREADS;            // f32x4 of weights per thread per chunk (6 or 8 / 3 or 4)
    constexpr int RPW = KCG / (NTHREADS / 64);             // activation rows per wave per chunk (4)
    constexpr int NJ = MT == 128 ? 3 : 5;                  // 64-lane sweeps per activation row (span <= 192 / 320)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int V = p.V;
    float *Wl = smem;                                      // [R][KCG][MT]
    float *Bx = smem + R * KCG * MT;                       // [KCG][ldb]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    const int l31 = lane & 31, kh = lane >> 5;
    const unsigned wid = xcd_contiguous_id(blockIdx.x, gridDim.x);
    const int m0 = (int)(wid % p.mtiles) * MT, q0 = (int)((wid / p.mtiles) % p.qtiles) * NT;
    const int seg = (int)(wid / (p.mtiles * p.qtiles));
    const int Q = p.frames * V;
    const int qend = min(q0 + NT, Q);
    const int ta = div_magic(q0, p.vmagic), tb = div_magic(qend - 1, p.vmagic);
    const int span = (tb - ta + 1) * V;

    // per-lane adjacency entries of the two output columns this lane feeds (B operand: column = lane & 31)
    int eoff[2][6];
    float eval[2][6];
    int ioff[2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int q = min(q0 + wn * 64 + ni * 32 + l31, Q - 1);
        const int t = div_magic(q, p.vmagic);
        const int w = q - t * V, fb = (t - ta) * V;
        ioff[ni] = fb + w;
#pragma unroll
        for (int e = 0; e < 6; ++e) {
            const int r = e < 2 ? e : 2, k = e < 2 ? 0 : e - 2;          // subsets 0,1: one entry; subset 2: four
            const bool have = k < p.ell_cnt[r];
            const int idx = (r * V + w) * p.ell_w + min(k, p.ell_w - 1);
            eoff[ni][e] = fb + (have ? p.ell_src[idx] : 0);
            eval[ni][e] = have ? p.ell_val[idx] : 0.f;
        }
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[a][b][g] = 0.f;

    const float *seg_base = p.x + (int64_t)seg * p.x_seg_stride;
    const float *wbase = p.w + m0;
    // staging registers + chunk-invariant offsets
    f32x4 wv[WB];
    unsigned wgo[WB], wlo[WB];
#pragma unroll
    for (int u = 0; u < WB; ++u) {
        const int e = u * NTHREADS + tid;                  // exact cover: R*KCG*M4 is a multiple of 256
        const int row = e / M4, m4 = e % M4;
        wgo[u] = (unsigned)(((row / KCG) * p.CinPad + (row % KCG)) * p.Mpad + m4 * 4);
        wlo[u] = (unsigned)(e * 4);
    }
    float bv[RPW][NJ];
    unsigned bgo[NJ], blo[NJ];
#pragma unroll
    for (int u = 0; u < NJ; ++u) {
        const int j = min(u * 64 + lane, span - 1);
        bgo[u] = (unsigned)(ta * V + j);                   // always inside [0, Q): whole frames of this segment
        blo[u] = (unsigned)j;
    }
    auto issue_w = [&](int c0) {
        const float *wc = wbase + (size_t)c0 * p.Mpad;
#pragma unroll
        for (int u = 0; u < WB; ++u) wv[u] = *reinterpret_cast<const f32x4 *>(wc + wgo[u]);
    };
    auto issue_x = [&](int c0) {
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            const int c = c0 + wave + rr * (NTHREADS / 64);
            const float *src = seg_base + (int64_t)min(c, p.Cin - 1) * p.x_chan_stride;
            const float m = c < p.Cin ? 1.f : 0.f;
#pragma unroll
            for (int u = 0; u < NJ; ++u) bv[rr][u] = src[bgo[u]] * m;
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int u = 0; u < WB; ++u) *reinterpret_cast<f32x4 *>(Wl + wlo[u]) = wv[u];
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            float *dst = Bx + (wave + rr * (NTHREADS / 64)) * p.ldb;
#pragma unroll
            for (int u = 0; u < NJ; ++u) dst[blo[u]] = bv[rr][u];
        }
    };

    const int offA = wm * 64 + l31;
    const int cpad = p.CinPad;                             // multiple of CSK_CPAD == KCG (zero-padded weights)
    auto mfma_steps = [&](int s_begin, int s_end) {
#pragma unroll 2
        for (int s = s_begin; s < s_end; ++s) {
            const int kk = 2 * s + kh;
            const float *bx = Bx + kk * p.ldb;
            float b[R][2];
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const float x0 = bx[eoff[ni][0]];
                b[0][ni] = eval[ni][0] * x0;
                b[1][ni] = eval[ni][1] * bx[eoff[ni][1]];
                float s2 = eval[ni][2] * bx[eoff[ni][2]];
                s2 = fmaf(eval[ni][3], bx[eoff[ni][3]], s2);
                s2 = fmaf(eval[ni][4], bx[eoff[ni][4]], s2);
                s2 = fmaf(eval[ni][5], bx[eoff[ni][5]], s2);
                b[2][ni] = s2;
                if (CONVRES) b[R - 1][ni] = bx[ioff[ni]];
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float *wr = Wl + (r * KCG + kk) * MT + offA;
                const float a0 = wr[0], a1 = wr[32];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b[r][0], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b[r][1], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b[r][0], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b[r][1], acc[1][1], 0, 0, 0);
            }
        }
    };
    issue_w(0);
    issue_x(0);
    int c0 = 0;
    for (; c0 + KCG < cpad; c0 += KCG) {
        __syncthreads();
        commit();
        __syncthreads();
        issue_w(c0 + KCG);                                 // next chunk's loads fly underneath the MFMAs
        issue_x(c0 + KCG);
        __builtin_amdgcn_s_setprio(1);
        mfma_steps(0, KCG / 2);
        __builtin_amdgcn_s_setprio(0);
    }
    __syncthreads();                                       // peeled last chunk: the staging registers are dead,
    commit();                                              // so the epilogue operands are loaded under its MFMAs
    __syncthreads();
    float bb[2][16], rv[2][2][16];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int g = 0; g < 16; ++g) bb[mi][g] = p.bias[m0 + wm * 64 + mi * 32 + 4 * kh + (g & 3) + 8 * (g >> 2)];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int qc = min(q0 + wn * 64 + ni * 32 + l31, Q - 1);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int co = m0 + wm * 64 + mi * 32 + 4 * kh + (g & 3) + 8 * (g >> 2);
                rv[ni][mi][g] = CONVRES ? 0.f : seg_base[(int64_t)min(co, p.Cout - 1) * p.x_chan_stride + qc];
            }
    }
    mfma_steps(0, KCG / 2);

    // epilogue: ReLU(acc + bias + identity residual); permlane32_swap pairs the ni = 0/1 registers so that every
    // store instruction writes one 256-B contiguous row segment (see tcn_stage_kernel)
    float *oseg = p.y + (int64_t)seg * p.y_seg_stride;
    const int qb = q0 + wn * 64 + lane;
    const bool qv = qb < Q;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int row0 = m0 + wm * 64 + mi * 32 + (g & 3) + 8 * (g >> 2);
            const float v0 = fmaxf(acc[mi][0][g] + bb[mi][g] + rv[0][mi][g], 0.f);
            const float v1 = fmaxf(acc[mi][1][g] + bb[mi][g] + rv[1][mi][g], 0.f);
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v0), __float_as_uint(v1), false, false);
            if (qv && row0 < p.Cout) oseg[(int64_t)row0 * p.y_chan_stride + qb] = __uint_as_float(sw[0]);
            if (qv && row0 + 4 < p.Cout) oseg[(int64_t)(row0 + 4) * p.y_chan_stride + qb] = __uint_as_float(sw[1]);
        }
}

// ------------------------------------------------------------------------------------------------
// A-GCN attention (models/a_gcn/a_gcn.py:53-63): per sample n and subset i
//     logits[v, w] = sum_{k,t} Ea[i][k][t][v] * Eb[i][k][t][w] / (inter * T)
//     adj[i][v, w] = softmax over v (dim -2) of logits + (A + graph_attn)[i][v, w]
// E = (n_seg, 6*inter, T, V): channels [i*inter + k] = a_conv_i, [3*inter + i*inter + k] = b_conv_i (biases
// included), produced by csk_tcn_stage_f32 as a 1x1 conv.  Output: the column-wise dense ELL values
// ell_val[n][i][w][v] = adj[i][v, w] consumed by gcn_stage_kernel with adj_seg_stride = 3*V*V.
// One workgroup per (n, i); rows (k,t) are streamed through LDS, thread p owns pairs (v,w) = p, p+256, p+512.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void agcn_attention_kernel(const float *__restrict__ E, const float *__restrict__ a_sum,
                                                             float *__restrict__ ell_val, int inter, int T, int V,
                                                             int64_t e_seg_stride, int64_t e_chan_stride) {
    // logits = Ea^T . Eb over K = inter*T rows as an fp32-MFMA product: A[i = v][k] = Ea[row k][v],
    // B[k][j = w] = Eb[row k][w] (V <= 32 columns used), operands straight from global memory (each row is V
    // contiguous floats).  The four waves take interleaved k-steps (2 rows each) and their partial 32x32 tiles
    // are summed through LDS.
    __shared__ float part[4][32][33];
    const int n = blockIdx.x, i = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5;
    const int K = inter * T;
    const float *ea = E + (int64_t)n * e_seg_stride + (int64_t)i * inter * e_chan_stride;
    const float *eb = E + (int64_t)n * e_seg_stride + (int64_t)(3 + i) * inter * e_chan_stride;
    f32x16 acc;
#pragma unroll
    for (int g = 0; g < 16; ++g) acc[g] = 0.f;
    // this lane's row index g = 2 * (4 s + wave) + kh, tracked as (channel kc, time t) without divisions
    int g = 2 * wave + kh;
    int kc = g / T, t = g - kc * T;
    const bool col = l31 < V;
    constexpr int UN = 4;
    for (; g < K + 8 * UN; g += 8 * UN) {
        float av[UN], bv[UN];
        int gg = g, kk = kc, tt = t;
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const bool ok = col && gg < K;
            const int64_t off = (int64_t)min(kk, inter - 1) * e_chan_stride + (int64_t)tt * V + min(l31, V - 1);
            const float xa = ea[off], xb = eb[off];
            av[u] = ok ? xa : 0.f;
            bv[u] = ok ? xb : 0.f;
            gg += 8; tt += 8;
            while (tt >= T) { tt -= T; ++kk; }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
        kc = kk; t = tt;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) part[wave][(r & 3) + 8 * (r >> 2) + 4 * kh][l31] = acc[r];   // [v][w]
    __syncthreads();
    if (tid < V) {                                // softmax over v (dim -2) for column w = tid
        const int w = tid;
        float lg[32];
        float m = -INFINITY;
        for (int v = 0; v < V; ++v) {
            lg[v] = (part[0][v][w] + part[1][v][w] + part[2][v][w] + part[3][v][w]) / (float)K;
            m = fmaxf(m, lg[v]);
        }
        float sum = 0.f;
        for (int v = 0; v < V; ++v) sum += expf(lg[v] - m);
        float *dst = ell_val + ((int64_t)(n * 3 + i) * V + w) * V;
        for (int v = 0; v < V; ++v) dst[v] = expf(lg[v] - m) / sum + a_sum[(i * V + v) * V + w];
    }
}

// ------------------------------------------------------------------------------------------------
// pre / post
// ------------------------------------------------------------------------------------------------
__global__ void input_norm_kernel(const float *__restrict__ x, const float *__restrict__ scale,
                                  const float *__restrict__ shift, float *__restrict__ h, int C, int T, int V, int M,
                                  int64_t h_seg_stride, int64_t h_chan_stride, int64_t total) {
    // one thread per input element, x index = (((n*C + c)*T + t)*V + v)*M + m  (reads fully coalesced)
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = i;
        const int m = r % M; r /= M;
        const int v = r % V; r /= V;
        const int t = r % T; r /= T;
        const int c = r % C;
        const int64_t n = r / C;
        const int ch = (m * V + v) * C + c;
        h[(n * M + m) * h_seg_stride + (int64_t)c * h_chan_stride + (int64_t)t * V + v] = fmaf(x[i], scale[ch], shift[ch]);
    }
}

// feat[n, c] = mean over m and over TV positions; one wave per (n, c)
__global__ __launch_bounds__(256) void pool_kernel(const float *__restrict__ h, float *__restrict__ feat, int N, int M,
                                                   int C, int TV, float scale) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;   // (n, c)
    if (row >= (int64_t)N * C) return;
    const int n = row / C, c = row % C;
    float s = 0.f;
    for (int m = 0; m < M; ++m) {
        const float *src = h + ((int64_t)(n * M + m) * C + c) * TV;
        for (int j = lane; j < TV; j += 64) s += src[j];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) feat[row] = s / (float)((int64_t)M * TV) * scale;
}

// logits[n, k] = feat[n] . fc_w[k] + fc_b[k]; one wave per output
__global__ __launch_bounds__(256) void fc_kernel(const float *__restrict__ feat, const float *__restrict__ w,
                                                 const float *__restrict__ b, float *__restrict__ logits, int N, int C,
                                                 int classes) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t o = (int64_t)blockIdx.x * 4 + wave;
    if (o >= (int64_t)N * classes) return;
    const int n = o / classes, k = o % classes;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s = fmaf(feat[(int64_t)n * C + c], w[(int64_t)k * C + c], s);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) logits[o] = s + b[k];
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------

// diagnostics, read from the environment per call only when CSK_DIAG is set at library load (so that the normal
// launch path never touches the environment): CSK_STAMPS=<device ptr> enables the s_memtime stamps of
// tcn_stage_kernel (tools/stamp_probe.py), CSK_GCN_GENERAL=1 forces the general (dense-capable) GCN kernel.
static const bool g_diag = getenv("CSK_DIAG") != nullptr;
static unsigned long long *diag_stamps() {
    if (!g_diag) return nullptr;
    const char *d = getenv("CSK_STAMPS");
    return d ? (unsigned long long *)strtoull(d, nullptr, 0) : nullptr;
}
static bool force_general_gcn() { return g_diag && getenv("CSK_GCN_GENERAL") != nullptr; }

// pick the <MT, NJ> instantiation, raise its dynamic-LDS cap, launch
template <typename P, typename K>
static int launch_stage(bool big, bool small_span, dim3 grid, size_t lds, hipStream_t s, const P &p, K k128a, K k128b,
                        K k64a, K k64b) {
    K k = big ? (small_span ? k128a : k128b) : (small_span ? k64a : k64b);
    if (const int e = csk_ensure_lds((const void *)k, lds)) return e;
    hipLaunchKernelGGL(k, grid, dim3(NTHREADS), lds, s, p);
    return (int)hipGetLastError();
}

extern "C" int csk_tcn_stage_f32(const float *y, const float *w, const float *x_res, const float *w_res,
                                 const float *bias, float *out, int n_seg, int c, int c_out, int t_in, int V, int k,
                                 int stride, int pad, int res_mode, int c_res, int t_res, int res_off, int relu,
                                 void *stream) {
    if (!y || !w || !bias || !out) CSK_FAIL("tcn_stage: null pointer");
    if (n_seg <= 0 || c <= 0 || c_out <= 0 || t_in <= 0 || V < 2 || V > 64) CSK_FAIL("tcn_stage: bad dims");
    if (k < 1 || k > 9 || stride < 1 || pad < 0 || pad >= k) CSK_FAIL("tcn_stage: bad k/stride/pad (k <= 9)");
    if (t_in + 2 * pad < k) CSK_FAIL("tcn_stage: t_in too short for kernel");
    const int t_out = (t_in + 2 * pad - k) / stride + 1;
    if (res_mode != CSK_RES_NONE) {
        if (!x_res) CSK_FAIL("tcn_stage: residual requested without x_res");
        if (res_mode == CSK_RES_IDENTITY && c_res != c_out) CSK_FAIL("tcn_stage: identity residual needs c_res == c_out");
        if (res_mode == CSK_RES_CONV && !w_res) CSK_FAIL("tcn_stage: conv residual without w_res");
        if ((t_out - 1) * stride + res_off >= t_res || res_off < 0) CSK_FAIL("tcn_stage: residual frames out of range");
    }
    if ((int64_t)t_in * V >= (1 << 26)) CSK_FAIL("tcn_stage: T*V too large for 32-bit position arithmetic");
    TcnParams p;
    p.y = y; p.w = w; p.xres = x_res ? x_res : y; p.wres = w_res; p.bias = bias; p.out = out;
    p.C = c; p.Cpad = round_up(c, CSK_CPAD); p.Cout = c_out; p.Mpad = round_up(c_out, CSK_MT);
    p.Tin = t_in; p.Tout = t_out; p.V = V; p.K = k; p.stride = stride; p.pad = pad;
    p.res_mode = res_mode; p.Cres = c_res > 0 ? c_res : 1; p.CresPad = round_up(p.Cres, CSK_CPAD);
    p.Tres = t_res > 0 ? t_res : 1; p.res_off = res_off; p.relu = relu; p.vmagic = vmagic_of(V);
    p.stamps = diag_stamps();
    p.prio = !(g_diag && getenv("CSK_NOPRIO"));
    const bool big = (p.Mpad % 128) == 0;
    const int MT = big ? 128 : 64, NT = 16384 / MT;
    const int max_dt = (NT + V - 2) / V;
    p.ldb = round_up((stride * max_dt + k) * V, 4);
    const size_t lds = (size_t)(k * KC * MT + KC * p.ldb) * sizeof(float);
    if (lds > 160 * 1024) CSK_FAIL("tcn_stage: LDS tile %zu B exceeds 160 KiB", lds);
    const int Q = t_out * V;
    p.qtiles = (Q + NT - 1) / NT; p.mtiles = p.Mpad / MT;
    if ((int64_t)p.qtiles * p.mtiles * n_seg >= (1ll << 31)) CSK_FAIL("tcn_stage: grid too large");
    dim3 grid(p.qtiles * p.mtiles * n_seg);
    const int nj = (p.ldb + 63) / 64;
    if (nj > 14) CSK_FAIL("tcn_stage: activation tile of %d positions per channel exceeds the staged maximum (896)", p.ldb);
    return launch_stage(big, nj <= 9, grid, lds, (hipStream_t)stream, p,
                        tcn_stage_kernel<128, 9>, tcn_stage_kernel<128, 14>, tcn_stage_kernel<64, 9>, tcn_stage_kernel<64, 14>);
}

extern "C" int csk_gcn_stage_f32(const float *x, float *y, const float *w, const float *bias, const int32_t *ell_src,
                                 const float *ell_val, const int32_t *ell_cnt, int ell_w, int64_t adj_seg_stride,
                                 int adj_per_frame,
                                 int n_seg, int c_in, int c_out, int frames, int V, int64_t x_seg_stride,
                                 int64_t x_chan_stride, int64_t y_seg_stride, int64_t y_chan_stride, int res_mode,
                                 void *stream) {
    if (!x || !y || !w || !bias || !ell_src || !ell_val || !ell_cnt) CSK_FAIL("gcn_stage: null pointer");
    if (n_seg <= 0 || c_in <= 0 || c_out <= 0 || frames <= 0 || V < 2 || V > 64) CSK_FAIL("gcn_stage: bad dims");
    if (ell_w < 1 || ell_w > V) CSK_FAIL("gcn_stage: ell_w must be in [1, V]");
    if (res_mode != CSK_RES_IDENTITY && res_mode != CSK_RES_CONV) CSK_FAIL("gcn_stage: res_mode must be identity or conv");
    if (res_mode == CSK_RES_IDENTITY && c_in != c_out) CSK_FAIL("gcn_stage: identity residual needs c_in == c_out");
    GcnParams p;
    p.x = x; p.w = w; p.bias = bias; p.y = y; p.ell_src = ell_src; p.ell_val = ell_val;
    for (int i = 0; i < 3; ++i) {
        if (ell_cnt[i] < 0 || ell_cnt[i] > ell_w) CSK_FAIL("gcn_stage: ell_cnt[%d] out of range", i);
        p.ell_cnt[i] = ell_cnt[i];
    }
    p.ell_w = ell_w; p.adj_seg_stride = adj_seg_stride;
    p.x_seg_stride = x_seg_stride; p.x_chan_stride = x_chan_stride;
    p.y_seg_stride = y_seg_stride; p.y_chan_stride = y_chan_stride;
    p.Cin = c_in; p.CinPad = round_up(c_in, CSK_CPAD); p.Cout = c_out; p.Mpad = round_up(c_out, CSK_MT);
    p.frames = frames; p.V = V; p.R = res_mode == CSK_RES_CONV ? 4 : 3; p.res_mode = res_mode;
    // per-segment adjacencies are dense by contract (include/cskel.h): ell_w == V, ell_cnt == {V,V,V}, src[e] == e
    p.dense = adj_seg_stride != 0 && ell_w == V && ell_cnt[0] == V && ell_cnt[1] == V && ell_cnt[2] == V;
    p.adj_per_frame = adj_per_frame != 0;
    if (p.adj_per_frame && !p.dense) CSK_FAIL("gcn_stage: per-frame adjacency must be dense (ell_w == V, ell_cnt == V)");
    p.vmagic = vmagic_of(V);
    const bool big = (p.Mpad % 128) == 0;
    const int MT = big ? 128 : 64, NT = 16384 / MT;
    const int max_dt = (NT + V - 2) / V;
    p.ldb = round_up((max_dt + 1) * V, 4);
    p.lds_frames = max_dt + 1;
    const size_t lds = (size_t)(p.R * KC * MT + p.R * KC * NT + KC * p.ldb + 3 * V * ell_w * (1 + (p.adj_per_frame ? p.lds_frames : 1))) * sizeof(float);
    if (lds > 160 * 1024) CSK_FAIL("gcn_stage: LDS tile %zu B exceeds 160 KiB", lds);
    const int Q = frames * V;
    if ((int64_t)frames * V >= (1 << 26)) CSK_FAIL("gcn_stage: frames*V too large for 32-bit position arithmetic");
    p.qtiles = (Q + NT - 1) / NT; p.mtiles = p.Mpad / MT;
    if ((int64_t)p.qtiles * p.mtiles * n_seg >= (1ll << 31)) CSK_FAIL("gcn_stage: grid too large");
    dim3 grid(p.qtiles * p.mtiles * n_seg);
    // sparse-graph fast path: shared adjacency with <= 1/1/4 non-zeros per column, activation tile <= 320 positions
    const bool sparse = adj_seg_stride == 0 && ell_cnt[0] <= 1 && ell_cnt[1] <= 1 && ell_cnt[2] <= 4 && p.ldb <= (big ? 192 : 320) &&
                        !force_general_gcn();
    if (sparse) {
        const int R = p.R;
        const size_t lds2 = (size_t)(R * KCG * MT + KCG * p.ldb) * sizeof(float);
        void (*k)(GcnParams) = big ? (R == 4 ? gcn_stage_sparse_kernel<128, true> : gcn_stage_sparse_kernel<128, false>)
                                   : (R == 4 ? gcn_stage_sparse_kernel<64, true> : gcn_stage_sparse_kernel<64, false>);
        const int e = csk_ensure_lds((const void *)k, lds2);
        if (e) return e;
        hipLaunchKernelGGL(k, grid, dim3(NTHREADS), lds2, (hipStream_t)stream, p);
        return (int)hipGetLastError();
    }
    const int nj = (p.ldb + 63) / 64;
    if (nj > 14) CSK_FAIL("gcn_stage: activation tile of %d positions per channel exceeds the staged maximum (896)", p.ldb);
    return launch_stage(big, nj <= 9, grid, lds, (hipStream_t)stream, p,
                        gcn_stage_kernel<128, 9>, gcn_stage_kernel<128, 14>, gcn_stage_kernel<64, 9>, gcn_stage_kernel<64, 14>);
}

extern "C" int csk_input_norm_f32(const float *x, const float *scale, const float *shift, float *h, int N, int C,
                                  int T, int V, int M, int64_t h_seg_stride, int64_t h_chan_stride, void *stream) {
    if (!x || !scale || !shift || !h) CSK_FAIL("input_norm: null pointer");
    if (N <= 0 || C <= 0 || T <= 0 || V <= 0 || M <= 0) CSK_FAIL("input_norm: bad dims");
    const int64_t total = (int64_t)N * C * T * V * M;
    const int64_t want = (total + 255) / 256;
    const int blocks = (int)(want < 8192 ? want : 8192);
    hipLaunchKernelGGL(input_norm_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, scale, shift, h, C, T, V,
                       M, h_seg_stride, h_chan_stride, total);
    return (int)hipGetLastError();
}

extern "C" int csk_fc_f32(const float *feat, const float *fc_w, const float *fc_b, float *logits, int N, int C,
                          int classes, void *stream) {
    if (!feat || !fc_w || !fc_b || !logits) CSK_FAIL("fc: null pointer");
    if (N <= 0 || C <= 0 || classes <= 0) CSK_FAIL("fc: bad dims");
    const int64_t outs = (int64_t)N * classes;
    hipLaunchKernelGGL(fc_kernel, dim3((unsigned)((outs + 3) / 4)), dim3(256), 0, (hipStream_t)stream, feat, fc_w, fc_b,
                       logits, N, C, classes);
    return (int)hipGetLastError();
}

extern "C" int csk_pool_fc_f32(const float *h, const float *fc_w, const float *fc_b, float *feat, float *logits, int N,
                               int M, int C, int TV, int classes, void *stream) {
    if (!h || !feat) CSK_FAIL("pool_fc: null pointer");
    if (N <= 0 || M <= 0 || C <= 0 || TV <= 0) CSK_FAIL("pool_fc: bad dims");
    const int64_t rows = (int64_t)N * C;
    hipLaunchKernelGGL(pool_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, h, feat, N, M,
                       C, TV, 1.0f);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    if (logits) return csk_fc_f32(feat, fc_w, fc_b, logits, N, C, classes, stream);
    return 0;
}

extern "C" int csk_agcn_attention_f32(const float *E, const float *a_sum, float *ell_val, int n_seg, int inter, int T,
                                      int V, int64_t e_seg_stride, int64_t e_chan_stride, void *stream) {
    if (!E || !a_sum || !ell_val) CSK_FAIL("agcn_attention: null pointer");
    if (n_seg <= 0 || inter <= 0 || T <= 0 || V < 2 || V > 32) CSK_FAIL("agcn_attention: bad dims (V <= 32)");
    hipLaunchKernelGGL(agcn_attention_kernel, dim3(n_seg, 3), dim3(256), 0, (hipStream_t)stream, E, a_sum, ell_val,
                       inter, T, V, e_seg_stride, e_chan_stride);
    return (int)hipGetLastError();
}

extern "C" int csk_pool_scaled_f32(const float *h, float *feat, int N, int M, int C, int TV, float scale, void *stream) {
    if (!h || !feat) CSK_FAIL("pool_scaled: null pointer");
    if (N <= 0 || M <= 0 || C <= 0 || TV <= 0) CSK_FAIL("pool_scaled: bad dims");
    const int64_t rows = (int64_t)N * C;
    hipLaunchKernelGGL(pool_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, h, feat, N, M,
                       C, TV, scale);
    return (int)hipGetLastError();
}
