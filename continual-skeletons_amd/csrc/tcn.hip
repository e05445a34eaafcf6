// tcn.hip -- TCN stage of the ST-GCN clip forward path (gfx950 / MI355X).  The GCN stage is in gcn.hip.
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
#include <type_traits>

#include "mfma_core.h"
#include "tcn_params.h"

// ------------------------------------------------------------------------------------------------
// TCN stage
// ------------------------------------------------------------------------------------------------

// VT: joints per frame as a compile-time constant (25 / 18: the tap shift of an LDS read becomes an immediate offset), 0 = run time
template <int MT, int NJ, bool K9 = false, bool SPLIT = false, int VT = 0>
__global__ __launch_bounds__(NTHREADS, 2) void tcn_stage_kernel(const TcnParams p) {
    constexpr int RES_G = VT ? 4 : 2;   // 8-channel groups per chunk of the conv-residual phase
    constexpr int OCC = 2;   // 3 (epilogue operands loaded after the K loop, <= 168 registers) was measured: the K loop's
                             // staging registers then spill and the stage runs 13-19 % slower
    constexpr int NT = 16384 / MT;
    constexpr int WM = MT / 64;
    constexpr int LAT_AHEAD = SPLIT ? 2 : 0;   // operand reads two k-steps ahead where a workgroup has its CU to itself (mfma_core.h)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Wl = smem;
    float *Bl = smem + p.K * KC * MT;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    const int l31 = lane & 31, kh = lane >> 5;
    // work item -> (m-tile fastest: shares the activation tile; then position tile: shares halos; then segment)
    const unsigned wid = xcd_contiguous_id(blockIdx.x, gridDim.x);
    const int m0 = (int)(wid % p.mtiles) * MT, q0 = (int)((wid / p.mtiles) % p.qtiles) * p.nt;
    const int segks = (int)(wid / (p.mtiles * p.qtiles));
    // SPLIT (few-tile launches, csk_tcn_stage_splitk_f32): this workgroup walks the channel range [cb, cb + cper) of its tile
    // and stores raw partial sums; bias, identity residual and ReLU belong to tcn_reduce_kernel
    const int seg = SPLIT ? segks / p.ksplit : segks, ks = SPLIT ? segks % p.ksplit : 0;
    const int cb = SPLIT ? ks * p.cper : 0;
    const int Cl = SPLIT ? min(p.C - cb, p.cper) : p.C;               // real channels of the range (>= 1 by construction)
    const int CpadL = SPLIT ? min(p.Cpad - cb, p.cper) : p.Cpad;
    const int V = VT ? VT : p.V, Q = p.Tout * V;
    const int qend = min(q0 + p.nt, Q);          // p.nt == NT unless the staged span had to be narrowed (large stride * V)
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
    unsigned long long st0 = 0, st1 = 0, st2 = 0;
    if (p.stamps) st0 = __builtin_amdgcn_s_memtime();
    WStage<MT> ws;
    BStage<NJ> bs;
    // epilogue operands: 32 biases + (identity residual) 64 block-input values per lane.  They are loaded
    // UNDER THE LAST CHUNK'S MFMAs (the K loops are peeled by one iteration; the staging registers are dead
    // there), unconditionally (clamped indices, bias padded to Mpad), so the epilogue itself is stores only.
    float *oseg = SPLIT ? p.part + (int64_t)segks * p.Cout * Q : p.out + (int64_t)seg * p.Cout * Q;
    const float *rseg = p.xres + (int64_t)seg * p.Cres * p.Tres * V;
    const int64_t rcs = (int64_t)p.Tres * V;
    const bool ident = !SPLIT && p.res_mode == CSK_RES_IDENTITY;
    float bv[2][16], rv[2][2][16];
    // Rows of this wave: rbase + mi*32 + (g & 3) + 8*(g >> 2) (+ 4*kh in the accumulator layout); everything but the
    // lane's own offset is wave-uniform.  On full tiles (all MT rows exist) the row base pointers are formed on the
    // scalar unit and each access carries one 32-bit lane byte offset (ld_lane / st_lane); the general form
    // (clamped rows, per-element predicates, ~13 instructions per access) is kept for ragged channel counts.
    const int rbase = m0 + wm * 64;
    const bool full = p.fast_epi && m0 + MT <= p.Cout;
    const unsigned kh4 = 4u * (unsigned)kh;
    auto load_half = [&](int mi) {
        if (full) {
#pragma unroll
            for (int g = 0; g < 16; ++g) bv[mi][g] = SPLIT ? 0.f : ld_lane(p.bias + (rbase + mi * 32 + (g & 3) + 8 * (g >> 2)), kh4 * 4u);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int qc = min(q0 + wn * 64 + ni * 32 + l31, qend - 1);
                const int t = div_magic(qc, p.vmagic);
                const unsigned qres = ident ? 4u * (kh4 * (unsigned)rcs + (unsigned)((t * p.stride + p.res_off) * V + (qc - t * V))) : 0u;
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const float *rrow = rseg + (int64_t)(rbase + mi * 32 + (g & 3) + 8 * (g >> 2)) * rcs;
                    rv[ni][mi][g] = ident ? ld_lane(rrow, qres) : 0.f;
                }
            }
        } else {
#pragma unroll
            for (int g = 0; g < 16; ++g) bv[mi][g] = SPLIT ? 0.f : p.bias[rbase + mi * 32 + 4 * kh + (g & 3) + 8 * (g >> 2)];
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int qc = min(q0 + wn * 64 + ni * 32 + l31, qend - 1);
                const int t = div_magic(qc, p.vmagic);
                const int qres = ident ? (t * p.stride + p.res_off) * V + (qc - t * V) : 0;
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const int co = rbase + mi * 32 + 4 * kh + (g & 3) + 8 * (g >> 2);
                    rv[ni][mi][g] = ident ? rseg[(int64_t)min(co, p.Cout - 1) * rcs + qres] : 0.f;
                }
            }
        }
    };
    const bool conv_res = p.res_mode == CSK_RES_CONV && ks == 0;       // split-K: the residual conv rides in split 0
    // ---- phase 1: k x 1 temporal conv over y.  The loop is written once over the activation-staging type: tiles whose
    // whole staged span lies inside the row (no zero padding to apply: all but the 1-2 tiles at either end of a
    // sequence) stage with 16-byte loads / LDS writes (BStage4), the others element-wise with clamp + select (BStage).
    {
        const int fa = p.stride * ta - p.pad;
        const int span = (p.stride * (tb - ta) + p.K) * V;
        const int64_t cs = (int64_t)p.Tin * V;
        const float *seg_base = p.y + (int64_t)seg * p.C * p.Tin * V + (int64_t)cb * cs;
        const float *wbase = p.w + m0 + (size_t)cb * p.Mpad;
        // 9-tap chunks: the compact weight staging (offsets of a thread's slots differ by wave-uniform constants)
        using WS9 = typename std::conditional<MT == 128, WStage9x128, WStage9x64>::type;
        typename std::conditional<K9, WS9, WStage<MT> &>::type ws1 = [&]() -> decltype(auto) {
            if constexpr (K9) { WS9 w9; w9.setup(p.Cpad, p.Mpad, tid); return w9; }
            else { ws.setup(p.K, p.Cpad, p.Mpad, tid); return (ws); }
        }();
        auto phase1 = [&](auto &bx) {
            // straight-line 3-tap MFMA segments (mfma_taps_ct); the element-wise boundary-tile path of the widest-span
            // instantiation keeps the rolled loop (register budget: it would spill 3 registers)
            constexpr bool CT = K9 && !(MT == 128 && NJ == 9 && std::is_same<typename std::remove_reference<decltype(bx)>::type, BStage<NJ>>::value);
            ws1.issue(wbase);
            bx.issue(seg_base, Cl, cs, 0, wave);
            int c0 = 0;
            unsigned long long ph0 = 0, ph1 = 0, ph2 = 0, ph3 = 0, tq = 0;   // diagnostic phase sums (p.stamps only)
            for (; c0 + KC < CpadL; c0 += KC) {
                if (p.stamps) tq = __builtin_amdgcn_s_memtime();
                __syncthreads();                       // previous chunk's LDS reads are done
                if (p.stamps) { const unsigned long long t = __builtin_amdgcn_s_memtime(); ph0 += t - tq; tq = t; }
                ws1.commit(Wl);
                bx.commit(Bl, p.ldb, wave);
                __syncthreads();
                if (p.stamps) { const unsigned long long t = __builtin_amdgcn_s_memtime(); ph1 += t - tq; tq = t; }
                if (p.stamps && c0 == 0) st1 = __builtin_amdgcn_s_memtime();
                {
                    // the next chunk's loads are issued in three bursts between three tap segments (see mfma_taps)
                    const float *wnext = wbase + (size_t)(c0 + KC) * p.Mpad;
                    const int cn = c0 + KC, t1 = (p.K + 2) / 3, t2 = min(p.K, 2 * t1);
                    if (p.stamps) { const unsigned long long t = __builtin_amdgcn_s_memtime(); ph2 += t - tq; tq = t; }
#pragma unroll
                    for (int j = 0; j < 3; ++j) ws1.issue_slot(j, wnext);
                    bx.template issue_third<0>(seg_base, Cl, cs, cn, wave);
                    // raised priority while in an MFMA segment: this wave then wins issue arbitration against the
                    // SIMD partner's commit / load-issue phase (+2 % measured)
                    if (p.prio) __builtin_amdgcn_s_setprio(1);
                    if (CT) mfma_taps_ct<MT, 3, LAT_AHEAD>(Wl, Bl, 0, p.ldb, V, offA, off[0], off[1], kh, acc);
                    else mfma_taps<MT>(Wl, Bl, 0, t1, p.ldb, V, offA, off[0], off[1], kh, acc);
                    __builtin_amdgcn_s_setprio(0);
#pragma unroll
                    for (int j = 3; j < 6; ++j) ws1.issue_slot(j, wnext);
                    bx.template issue_third<1>(seg_base, Cl, cs, cn, wave);
                    if (p.prio) __builtin_amdgcn_s_setprio(1);
                    if (CT) mfma_taps_ct<MT, 3, LAT_AHEAD>(Wl, Bl, 3, p.ldb, V, offA, off[0], off[1], kh, acc);
                    else if (t1 < t2) mfma_taps<MT>(Wl, Bl, t1, t2, p.ldb, V, offA, off[0], off[1], kh, acc);
                    __builtin_amdgcn_s_setprio(0);
#pragma unroll
                    for (int j = 6; j < 9; ++j) ws1.issue_slot(j, wnext);
                    bx.template issue_third<2>(seg_base, Cl, cs, cn, wave);
                    if (p.prio) __builtin_amdgcn_s_setprio(1);
                    if (CT) mfma_taps_ct<MT, 3, LAT_AHEAD>(Wl, Bl, 6, p.ldb, V, offA, off[0], off[1], kh, acc);
                    else if (t2 < p.K) mfma_taps<MT>(Wl, Bl, t2, p.K, p.ldb, V, offA, off[0], off[1], kh, acc);
                    __builtin_amdgcn_s_setprio(0);
                }
                if (p.stamps) { const unsigned long long t = __builtin_amdgcn_s_memtime(); ph3 += t - tq; tq = t; }
            }
            if (p.stamps && lane == 0) {
                unsigned long long *o = p.stamps + (size_t)gridDim.x * 6 + ((size_t)blockIdx.x * 4 + wave) * 4;
                o[0] = ph0; o[1] = ph1; o[2] = ph2; o[3] = ph3;
            }
            __syncthreads();                           // peeled last chunk
            ws1.commit(Wl);
            bx.commit(Bl, p.ldb, wave);
            __syncthreads();
        };
        const bool interior = p.vec_stage && fa >= 0 && fa * V + 4 * ((span + 3) / 4) <= p.Tin * V;     // uniform
        if (interior) {
            BStage4<(NJ + 3) / 4> b4;
            b4.setup(fa * V, span, lane);
            phase1(b4);
        } else {
            bs.setup(fa * V, span, p.Tin * V, lane);
            phase1(bs);
        }
        if (!conv_res && OCC == 2) { load_half(0); load_half(1); }
        if (K9 && NJ < 9 && !p.no_peel_ct) {   // the peeled last chunk in straight-line 3-tap segments too, where the register budget allows
                                               // (+0.1 ... +0.7 %; diagnostic CSK_TCN_NOPEELCT keeps it rolled)
            mfma_taps_ct<MT, 3, LAT_AHEAD>(Wl, Bl, 0, p.ldb, V, offA, off[0], off[1], kh, acc);
            mfma_taps_ct<MT, 3, LAT_AHEAD>(Wl, Bl, 3, p.ldb, V, offA, off[0], off[1], kh, acc);
            mfma_taps_ct<MT, 3, LAT_AHEAD>(Wl, Bl, 6, p.ldb, V, offA, off[0], off[1], kh, acc);
        } else {
            mfma_chunk<MT>(Wl, Bl, p.K, p.ldb, V, offA, off[0], off[1], kh, acc);
        }
    }
    // ---- phase 2: 1x1 strided residual conv over the block input (models/base.py:372-374).  A chunk holds G = 4 (or 2)
    // consecutive 8-channel groups laid out like the taps of a 9-tap chunk -- Wl [G][KC][MT], Bl [G * KC][ldb2], "tap" stride
    // KC * ldb2 -- so that the chunk's fixed cost (27 staging slots, two barriers) buys 16 G MFMAs per wave instead of 16: with
    // one group per chunk this phase took 14 % of a stride-2 tile for 1 / 9 of its MFMAs.  Channel pairs are accumulated in
    // the same order as before (bitwise the same sums).
    if (conv_res) {
        const int fa = p.stride * ta + p.res_off;
        const int span = (p.stride * (tb - ta) + 1) * V;
        const float *seg_base = p.xres + (int64_t)seg * p.Cres * p.Tres * V;
        const int64_t cs = (int64_t)p.Tres * V;
        const float *wbase = p.wres + m0;
        constexpr int NJ2 = VT ? (NJ * 64 - 8 * VT + 63) / 64 : NJ;        // span2 = span1 - 8 V <= 64 NJ - 8 V
        {
            // 4 groups per chunk in the instantiations with a compile-time joint count (their residual tile is NJ2 <= 6 sweeps
            // wide: 8 rows x 6 sweeps of staging registers); the others stage 2 groups -- the host sizes the LDS for it
            constexpr int G = RES_G;
            constexpr int WB2 = G * KC * (MT / 4) / NTHREADS;               // f32x4 of weights per thread and chunk (1 .. 4)
            static_assert(G * KC * (MT / 4) % NTHREADS == 0, "whole sweeps");
            float *Wl2 = smem, *Bl2 = smem + G * KC * MT;
            const int ldb2 = p.ldb2;
            // slot u of a thread: weight row u * RS + tid / (MT / 4) -- global and LDS offsets differ by wave-uniform constants
            constexpr int RS = NTHREADS / (MT / 4);                       // rows per sweep of 256 threads
            f32x4 wv[WB2];
            const unsigned wgo0 = (unsigned)((tid / (MT / 4)) * p.Mpad + (tid % (MT / 4)) * 4), wlo0 = (unsigned)(tid * 4);
            const unsigned wgs = (unsigned)(RS * p.Mpad);
            BStage<NJ2, 2 * G> b2;
            b2.setup(fa * V, span, p.Tres * V, lane);
            auto issue2 = [&](int c0) {
#pragma unroll
                for (int u = 0; u < WB2; ++u) wv[u] = *reinterpret_cast<const f32x4 *>(wbase + (size_t)c0 * p.Mpad + (size_t)u * wgs + wgo0);
                b2.issue(seg_base, p.Cres, cs, c0, wave);
            };
            auto commit2 = [&]() {
#pragma unroll
                for (int u = 0; u < WB2; ++u) *reinterpret_cast<f32x4 *>(Wl2 + u * (NTHREADS * 4) + wlo0) = wv[u];
                b2.commit(Bl2, ldb2, wave);
            };
            issue2(0);
            int c0 = 0;
            for (; c0 + G * KC < p.CresPad; c0 += G * KC) {
                __syncthreads();
                commit2();
                __syncthreads();
                issue2(c0 + G * KC);
                mfma_chunk<MT>(Wl2, Bl2, G, ldb2, KC * ldb2, offA, off[0], off[1], kh, acc);
            }
            __syncthreads();
            commit2();
            __syncthreads();
            mfma_chunk<MT>(Wl2, Bl2, G, ldb2, KC * ldb2, offA, off[0], off[1], kh, acc);
            // (the epilogue operands are loaded BEHIND this chunk, not under it: 96 registers beside 8 staged rows would spill;
            // one exposed load round trip per stride-2 tile of 200-400 us)
            __builtin_amdgcn_sched_barrier(0);
            if (OCC == 2) { load_half(0); load_half(1); }
        }
    }
    if (p.stamps) st2 = __builtin_amdgcn_s_memtime();
    // ---- epilogue: + bias (+ identity residual), ReLU, predicated stores.
    // C/D map: col = lane&31, row = (g&3) + 8(g>>2) + 4(lane>>5).  A plain store of accumulator register g
    // writes two 128-B half rows (rows r and r+4).  v_permlane32_swap of the ni=0 / ni=1 registers gives each
    // lane half the SAME row instead: lanes 0-31 columns 0-31, lanes 32-63 columns 32-63 of row r (first
    // result) and of row r+4 (second) -> every store instruction writes one 256-B contiguous row segment.
    const int qb = q0 + wn * 64 + lane;                    // column of this lane after the swap
    const bool qv = qb < qend;
    auto finish_half = [&](int mi) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            float v0 = acc[mi][0][g] + bv[mi][g] + rv[0][mi][g];
            float v1 = acc[mi][1][g] + bv[mi][g] + rv[1][mi][g];
            if (!SPLIT && p.relu) { v0 = relu_nan(v0); v1 = relu_nan(v1); }
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
    };
    if (OCC == 2) {
        finish_half(0);
        finish_half(1);
    } else {
        load_half(0);
        finish_half(0);
        load_half(1);
        finish_half(1);
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
// C ABI
// ------------------------------------------------------------------------------------------------
// split-K reduction of the TCN stage: out[seg][co][q] = ReLU?( sum_ks part[seg * ksplit + ks][co][q] (split order) + bias[co]
// + identity residual x_res[seg][co][(t * stride + res_off) * V + v] ); one thread per 4 consecutive (co, q) of a segment
__global__ __launch_bounds__(256) void tcn_reduce_kernel(const TcnParams p) {
    const int Q = p.Tout * p.V;
    const int64_t n = (int64_t)p.Cout * Q;
    const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    const int seg = blockIdx.y;
    if (i0 >= n) return;
    const float *pp = p.part + (int64_t)seg * p.ksplit * n;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    const bool vec = i0 + 4 <= n && (n & 3) == 0;
    for (int ks = 0; ks < p.ksplit; ++ks) {
        if (vec) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(pp + (int64_t)ks * n + i0);
            s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
        } else {
            for (int j = 0; j < 4; ++j) if (i0 + j < n) s[j] += pp[(int64_t)ks * n + i0 + j];
        }
    }
    float *o = p.out + (int64_t)seg * n;
    for (int j = 0; j < 4; ++j) {
        const int64_t i = i0 + j;
        if (i >= n) break;
        const int co = (int)(i / Q), q = (int)(i - (int64_t)co * Q);
        float v = s[j] + p.bias[co];
        if (p.res_mode == CSK_RES_IDENTITY) {
            const int t = div_magic(q, p.vmagic);
            v += p.xres[((int64_t)seg * p.Cres + co) * p.Tres * p.V + (int64_t)(t * p.stride + p.res_off) * p.V + (q - t * p.V)];
        }
        o[i] = p.relu ? relu_nan(v) : v;
    }
}

static int tcn_stage_impl(const float *y, const float *w, const float *x_res, const float *w_res,
                          const float *bias, float *out, int n_seg, int c, int c_out, int t_in, int V, int k,
                          int stride, int pad, int res_mode, int c_res, int t_res, int res_off, int relu, int ksplit,
                          float *partial, void *stream) {
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
    if (ksplit < 1 || ksplit > 64) CSK_FAIL("tcn_stage: ksplit must be in [1, 64]");
    if (ksplit > 1 && (!partial || ((uintptr_t)partial & 15))) CSK_FAIL("tcn_stage: split-K needs a 16-byte aligned partial-sum buffer");
    if (ksplit > 1 && k != 9) CSK_FAIL("tcn_stage: the split-K form is built for the 9-tap conv");
    TcnParams p;
    p.y = y; p.w = w; p.xres = x_res ? x_res : y; p.wres = w_res; p.bias = bias; p.out = out;
    p.C = c; p.Cpad = round_up(c, CSK_CPAD); p.Cout = c_out; p.Mpad = round_up(c_out, CSK_MT);
    p.Tin = t_in; p.Tout = t_out; p.V = V; p.K = k; p.stride = stride; p.pad = pad;
    p.res_mode = res_mode; p.Cres = c_res > 0 ? c_res : 1; p.CresPad = round_up(p.Cres, CSK_CPAD);
    p.Tres = t_res > 0 ? t_res : 1; p.res_off = res_off; p.relu = relu; p.vmagic = vmagic_of(V);
    // split-K: every range owns >= 1 real channel; fewer ranges than asked for if the channel count does not allow more
    p.cper = round_up((p.Cpad + ksplit - 1) / ksplit, KC);
    p.ksplit = ksplit > 1 ? (c + p.cper - 1) / p.cper : 1;
    p.part = partial;
    p.stamps = csk_diag_stamps();
    p.prio = !csk_diag_flag("CSK_NOPRIO");
    p.vec_stage = !csk_diag_flag("CSK_TCN_NOVEC");
    p.no_peel_ct = csk_diag_flag("CSK_TCN_NOPEELCT");
    // 32-bit lane byte offsets: 4 * (4 * row_stride + position) must stay below 2^32
    p.fast_epi = (int64_t)p.Tres * V < (1ll << 27) && (int64_t)t_out * V < (1ll << 27) && !csk_diag_flag("CSK_SLOW_EPI");
    // the register staging holds <= 9 (128-row tiles) / 14 (64-row tiles) x 64 positions of an activation row -- the
    // widest spill-free instantiations; tiles whose input span (stride * frames + k taps) * V is longer than that
    // (stride 2 with V > 32, stride 3, ...) are narrowed: a tile then covers nt < NT output positions and the remaining
    // MFMA columns idle.  Never the case for the ST-GCN shapes (V <= 25, stride <= 2).  Both tile heights are priced
    // (the 64-row kernel stages 14 sweeps: at V = 64, stride 2 it keeps 128 of its 256 columns where the 128-row kernel
    // keeps 1 of 128) and the better MFMA-column utilisation wins; a shape that loses more than half the columns either
    // way says so once on stderr.
    // the 16x16x4 tile family (tcn16.hip) for the shapes it is built for; bitwise the same sums (-2: not taken)
    if (p.ksplit == 1) {
        const int rc = csk_launch_tcn_stage16(p, n_seg, stream);
        if (rc != -2) return rc;
    }
    auto narrow = [&](int NT_, int nj_max_, int *ldb_) {
        int nt = NT_;
        for (;;) {
            const int max_dt = (nt + V - 2) / V;
            *ldb_ = round_up((stride * max_dt + k) * V, 4);
            if ((*ldb_ + 63) / 64 <= nj_max_ || nt == 1) return nt;
            nt = nt > 16 ? nt - 16 : nt > 1 ? nt / 2 : 1;
        }
    };
    bool big = (p.Mpad % 128) == 0;
    int ldb64 = 0, ldb128 = 0;
    const int nt64 = narrow(256, 14, &ldb64), nt128 = big ? narrow(128, 9, &ldb128) : 0;
    if (big && nt128 < 128 && nt64 * 128 > nt128 * 256) big = false;       // utilisation nt64 / 256 > nt128 / 128
    const int MT = big ? 128 : 64, NT = 16384 / MT;
    const int nj_max = big ? 9 : 14;
    p.nt = big ? nt128 : nt64;
    p.ldb = big ? ldb128 : ldb64;
    if (2 * p.nt < NT) {
        static bool warned = false;
        if (!warned) {
            warned = true;
            fprintf(stderr, "libcskel_hip: tcn_stage with V = %d, stride %d, k = %d keeps %d of %d tile columns (activation span "
                            "exceeds the staged maximum): correct, but far below the kernel's rate\n", V, stride, k, p.nt, NT);
        }
    }
    const int nj = (p.ldb + 63) / 64;
    if (nj > nj_max) CSK_FAIL("tcn_stage: activation tile of %d positions per channel exceeds the staged maximum (%d)", p.ldb, 64 * nj_max);
    size_t lds = (size_t)(k * KC * MT + KC * p.ldb) * sizeof(float);
    p.ldb2 = 4;
    bool vt_ok = true;                     // the compile-time-V instantiations stage FOUR 8-channel groups per residual chunk
    if (res_mode == CSK_RES_CONV) {
        // conv-residual phase: G 8-channel groups per chunk (CresPad is a multiple of 16: G = 2 always divides it)
        const int max_dt = (p.nt + V - 2) / V;
        p.ldb2 = round_up((stride * max_dt + 1) * V, 4);
        if (p.ldb2 > p.ldb) CSK_FAIL("tcn_stage: residual tile wider than the conv tile");
        auto lds2 = [&](int G) { return (size_t)(G * KC * MT + G * KC * p.ldb2) * sizeof(float); };
        vt_ok = p.CresPad % (4 * KC) == 0 && 2 * lds2(4) <= 160 * 1024;
        const size_t need = lds2(vt_ok && k == 9 && (V == 25 || V == 18) ? 4 : 2);
        if (need > lds) lds = need;
    }
    if (lds > 160 * 1024) CSK_FAIL("tcn_stage: LDS tile %zu B exceeds 160 KiB", lds);
    const int Q = t_out * V;
    p.qtiles = (Q + p.nt - 1) / p.nt; p.mtiles = p.Mpad / MT;
    if ((int64_t)p.qtiles * p.mtiles * n_seg * p.ksplit >= (1ll << 31)) CSK_FAIL("tcn_stage: grid too large");
    dim3 grid(p.qtiles * p.mtiles * n_seg * p.ksplit);
    // NJ = 64-lane sweeps of the activation staging per row: the smallest instantiation that covers the span (sweeps
    // beyond it re-load and re-commit its last position: wasted load / LDS-write slots).  128x128 tiles of the
    // stride-1 layers need 6 sweeps, not 9: -2.7 % on their tiles.
    void (*kern)(TcnParams) =
        big ? (nj <= 6 ? tcn_stage_kernel<128, 6> : tcn_stage_kernel<128, 9>)
            : (nj <= 6 ? tcn_stage_kernel<64, 6> : nj <= 9 ? tcn_stage_kernel<64, 9> : tcn_stage_kernel<64, 14>);
    if (k == 9 && !csk_diag_flag("CSK_TCN_NOK9"))     // the 9-tap form with straight-line 3-tap MFMA segments
        kern = big ? (nj <= 6 ? tcn_stage_kernel<128, 6, true> : tcn_stage_kernel<128, 9, true>)
                   : (nj <= 6 ? tcn_stage_kernel<64, 6, true> : nj <= 9 ? tcn_stage_kernel<64, 9, true> : tcn_stage_kernel<64, 14>);
    // the two skeleton sizes of the reference's datasets (NTU-25, OpenPose-18) with V as a compile-time constant: the tap
    // shift of an LDS read becomes an immediate offset (+0.1 ... +0.8 % per launch, in-process A/B CSK_TCN_NOVT)
    if (k == 9 && (V == 25 || V == 18) && vt_ok && !csk_diag_flag("CSK_TCN_NOK9") && !csk_diag_flag("CSK_TCN_NOVT") && nj <= 9) {
        if (V == 25)
            kern = big ? (nj <= 6 ? tcn_stage_kernel<128, 6, true, false, 25> : tcn_stage_kernel<128, 9, true, false, 25>)
                       : (nj <= 6 ? tcn_stage_kernel<64, 6, true, false, 25> : tcn_stage_kernel<64, 9, true, false, 25>);
        else
            kern = big ? (nj <= 6 ? tcn_stage_kernel<128, 6, true, false, 18> : tcn_stage_kernel<128, 9, true, false, 18>)
                       : (nj <= 6 ? tcn_stage_kernel<64, 6, true, false, 18> : tcn_stage_kernel<64, 9, true, false, 18>);
    }
    if (p.ksplit > 1 && (nj > 9 || n_seg > 65535)) {
        // the split-K instantiations stage <= 576 positions per channel and their reduction has the segment in grid.y: a request
        // outside that (a stride-2 64-row tile at V = 18: 702 positions; > 65 535 sequences) runs the plain kernel -- one
        // workgroup per tile walks the whole K loop, same result up to summation order, no partial sums, no reduction launch
        p.ksplit = 1;
        p.cper = p.Cpad;
        grid = dim3(p.qtiles * p.mtiles * n_seg);
    }
    if (p.ksplit > 1) {
        kern = big ? (nj <= 6 ? tcn_stage_kernel<128, 6, true, true> : tcn_stage_kernel<128, 9, true, true>)
                   : (nj <= 6 ? tcn_stage_kernel<64, 6, true, true> : tcn_stage_kernel<64, 9, true, true>);
    }
    if (const int e = csk_ensure_lds((const void *)kern, lds)) return e;
    hipLaunchKernelGGL(kern, grid, dim3(NTHREADS), lds, (hipStream_t)stream, p);
    if (p.ksplit > 1) {
        if (const int e = (int)hipGetLastError()) return e;
        const int64_t work = ((int64_t)c_out * Q + 3) / 4;
        hipLaunchKernelGGL(tcn_reduce_kernel, dim3((unsigned)((work + 255) / 256), n_seg), dim3(256), 0, (hipStream_t)stream, p);
    }
    return (int)hipGetLastError();
}

extern "C" int csk_tcn_stage_f32(const float *y, const float *w, const float *x_res, const float *w_res,
                                 const float *bias, float *out, int n_seg, int c, int c_out, int t_in, int V, int k,
                                 int stride, int pad, int res_mode, int c_res, int t_res, int res_off, int relu,
                                 void *stream) {
    return tcn_stage_impl(y, w, x_res, w_res, bias, out, n_seg, c, c_out, t_in, V, k, stride, pad, res_mode, c_res, t_res, res_off, relu,
                          1, nullptr, stream);
}

extern "C" int csk_tcn_stage_splitk_f32(const float *y, const float *w, const float *x_res, const float *w_res,
                                        const float *bias, float *out, int n_seg, int c, int c_out, int t_in, int V, int k,
                                        int stride, int pad, int res_mode, int c_res, int t_res, int res_off, int relu,
                                        int ksplit, float *partial, void *stream) {
    return tcn_stage_impl(y, w, x_res, w_res, bias, out, n_seg, c, c_out, t_in, V, k, stride, pad, res_mode, c_res, t_res, res_off, relu,
                          ksplit, partial, stream);
}

