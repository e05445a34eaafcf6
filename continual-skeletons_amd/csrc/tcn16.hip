// tcn16.hip -- the clip temporal conv (csk_tcn_stage_f32, models/base.py:279-304 + the block tail :376-387) on the 16x16x4 tile
// family of tile16.h: 64 output channels x 16 NB output positions per workgroup, NB = V = 25 (NTU) or 18 (OpenPose) column
// blocks = exactly 16 output frames; a wave owns 16 channels and all NB column blocks.
//
// Why.  In-kernel stamps of the 32x32x2 stage kernel: a C = 256 tile spends 590 k of its 712 k cycles in MFMAs (0.83 of the
// loop) -- one LDS round trip per 64-cycle MFMA pair, a 2 x 2 accumulator tile per wave, 144 MFMAs between two barriers.  The
// step kernels of round 6 (step16.hip) showed what the 16-wide block buys: a wave carries 25 independent accumulators, 450
// MFMAs between two barriers, and its K loop runs at 0.985 of the matrix pipe (tools/stamp16_probe.py).  This is the same
// tile for the clip layout: the temporal taps are address shifts of r * V positions in ONE staged row per input channel
// (16 output frames + 8 halo frames), so a fragment read is one VGPR base + an immediate.
//   stride 2: output frame t' reads input frames 2 t' + r: the row holds the 40 input frames of the tile in natural order and
//   the column -> row offset gains the term (c / V) * V -- one base register per column block instead of one for all.
// Stride 1: the same (8-channel chunk, tap, channel) order as tcn_stage_kernel and the same epilogue expression -- bitwise the same
// results (tests/test_gpu_clip_parity.py::test_tile_families_are_bitwise_interchangeable).  Stride 2 walks 4-channel chunks (its
// 39-frame rows and 25 base registers leave no room for 8-channel staging): the same sums in another fp32 order.  Which family
// a layer gets is a function of the layer alone (csk_launch_tcn_stage16), never of the batch.
// csk_launch_tcn_stage16 takes k = 9, V = 25 / 18, stride 1 / 2, pad * V a multiple of 4; the 32x32x2 kernels keep the rest
// (and the split-K / bf16x3 forms).
#include <type_traits>

#include "tile16.h"
#include "tcn_params.h"

namespace {

// Staging of KCHX channel rows x NQ quads of a CONTIGUOUS position range [ws, ws + 4 NQ) of a (C, L) tensor whose rows are
// L = row_len positions long: positions outside [0, L) read as zero (the conv's temporal zero padding / the tile past the
// end of the sequence).  ws is a multiple of 4 and never negative inside a quad that also holds valid positions, so a quad
// is in or out as a whole at the front; at the back a row whose length is not a multiple of 4 ends inside a quad: that quad
// is loaded from the last whole quad of the row and shifted (sh = its low goff bits).  Unit e = (channel kk, quad i), quad
// fastest; a thread owns units sweep * 256 + tid.
template <int KCHX, int NQ, int ROWX>
struct Clip16 {
    static constexpr int U = KCHX * NQ, NSW = (U + NTHREADS - 1) / NTHREADS;
    static_assert(KCHX * ROWX * 4 < 65536, "LDS byte offsets are packed two to a register");
    unsigned goff[NSW];            // byte offset of the (clamped) quad from (segment + c0 * row_len); bits [1:0] = shift sh
    unsigned loff2[(NSW + 1) / 2];
    unsigned vmask;                // bit u: the quad of sweep u holds at least one position inside the row
    f32x4 v[NSW];
    __device__ __forceinline__ void setup(int ws, int row_len, int tid) {
        vmask = 0;
#pragma unroll
        for (int u = 0; u < NSW; ++u) {
            const int e = min(u * NTHREADS + tid, U - 1);
            const int i = e % NQ, kk = e / NQ;
            const int pos = ws + 4 * i;
            const bool any = pos >= 0 && pos < row_len;
            const int pc = max(0, min(pos, row_len - 4));                // last whole quad of the row at most
            const int sh = any ? pos - pc : 0;                           // 0 except in the quad the row ends in
            goff[u] = (unsigned)((kk * row_len + pc) * 4) | (unsigned)sh;
            vmask |= any ? (1u << u) : 0u;
            const unsigned lo = (unsigned)(kk * ROWX + 4 * i) * 4u;
            if (u & 1) loff2[u / 2] |= lo << 16;
            else loff2[u / 2] = lo;
        }
    }
    // EDGE (wave-uniform): the tile's window reaches past an end of the row -> zero fill / shift; interior tiles load plainly
    template <int U0, int U1>
    __device__ __forceinline__ void issue_range(const float *__restrict__ base, bool edge) {
#pragma unroll
        for (int u = U0; u < U1; ++u) {
            unsigned g = goff[u];
            asm volatile("" : "+v"(g));
            const f32x4 x = *reinterpret_cast<const f32x4u16 *>(reinterpret_cast<const char *>(base) + (g & ~3u));
            if (edge) {
                const unsigned sh = g & 3u;
                const bool ok = (vmask >> u) & 1u;
                f32x4 r;
                r[0] = sh == 0 ? x[0] : sh == 1 ? x[1] : sh == 2 ? x[2] : x[3];
                r[1] = sh == 0 ? x[1] : sh == 1 ? x[2] : sh == 2 ? x[3] : 0.f;
                r[2] = sh == 0 ? x[2] : sh == 1 ? x[3] : 0.f;
                r[3] = sh == 0 ? x[3] : 0.f;
                v[u] = ok ? r : f32x4{0.f, 0.f, 0.f, 0.f};
            } else {
                v[u] = x;
            }
        }
    }
    template <int G>
    __device__ __forceinline__ void issue_third(const float *__restrict__ base, bool edge) { issue_range<G * NSW / 3, (G + 1) * NSW / 3>(base, edge); }
    __device__ __forceinline__ void issue(const float *__restrict__ base, bool edge) { issue_range<0, NSW>(base, edge); }
    __device__ __forceinline__ void commit(float *__restrict__ Bl) const {
#pragma unroll
        for (int u = 0; u < NSW; ++u) {
            const unsigned lo = (u & 1) ? loff2[u / 2] >> 16 : loff2[u / 2] & 0xffffu;
            *reinterpret_cast<f32x4 *>(reinterpret_cast<char *>(Bl) + lo) = v[u];
        }
    }
};

// Weights of a 9-tap, 8-channel chunk: 1152 quads = 4.5 per thread; slot u of a thread is row 16 u + tid / 16 of the (tap, channel)
// row space, i.e. tap 2 u + (tid / 128): global and LDS offsets of the slots differ by wave-uniform constants -- ONE offset
// register each (the WStage9x64 scheme of mfma_core.h); the upper half of the threads has no slot 4 and repeats its slot 3.
template <int LDW>
struct W16x9x8 {
    unsigned goff0, loff0, gstride, u4;
    f32x4 v[5];
    __device__ __forceinline__ void setup(int Cpad, int Mpad, int tid) {
        goff0 = (unsigned)((((tid >> 7) * Cpad + ((tid >> 4) & 7)) * Mpad + (tid & 15) * 4) * 4);
        loff0 = (unsigned)(((tid >> 4) * LDW + (tid & 15) * 4) * 4);
        gstride = (unsigned)(2 * Cpad * Mpad * 4);
        u4 = tid >= 128 ? 3u : 4u;
    }
    __device__ __forceinline__ void issue_one(int u, const float *__restrict__ base) {
        if (u > 4) return;
        unsigned g = goff0 + (u < 4 ? (unsigned)u : u4) * gstride;
        asm volatile("" : "+v"(g));
        v[u] = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(base) + g);
    }
    __device__ __forceinline__ void issue(const float *__restrict__ base) {
#pragma unroll
        for (int u = 0; u < 5; ++u) issue_one(u, base);
    }
    __device__ __forceinline__ void commit(float *__restrict__ Wl) const {
#pragma unroll
        for (int u = 0; u < 5; ++u)
            *reinterpret_cast<f32x4 *>(reinterpret_cast<char *>(Wl) + loff0 + (u < 4 ? (unsigned)u : u4) * (16u * LDW * 4u)) = v[u];
    }
};

// S = temporal stride (1 or 2), V = joints per frame = NB
template <int V, int S, int KCH_>
struct C16 {
    static constexpr int NB = V, NT = 16 * V, LDW = 80;
    static constexpr int KCH = KCH_;                             // channels per chunk (stride 2 stages 39 input frames per row: 4, register budget)
    static constexpr int NFR = S * 15 + 9;                       // input frames a tile reads (K = 9)
    static constexpr int SPAN = (NFR * V + 3) / 4 * 4, NQ = SPAN / 4;
    static constexpr int ROW = row16(SPAN);
    static constexpr int NFR_R = S * 15 + 1, SPAN_R = (NFR_R * V + 3) / 4 * 4, ROWR = row16(SPAN_R);   // conv-residual phase (1 tap)
    static constexpr int WSZ = 9 * KCH * LDW;
    static constexpr int LDS_FLOATS = imax(WSZ + KCH * ROW, KCH * LDW + KCH * ROWR);
};

template <int V, int S, int KCH_ = (S == 1 ? 8 : 4)>
__global__ __launch_bounds__(NTHREADS, 2) void tcn_stage16_kernel(const TcnParams p) {
    typedef C16<V, S, KCH_> G;
    constexpr int NB = G::NB, NT = G::NT, KCH = G::KCH, LDW = G::LDW, ROW = G::ROW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Wl = smem, *Bl = smem + G::WSZ;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kq = lane >> 4;
    // work item -> (m-tile fastest: shares the activation window; then position tile: shares halos; then segment)
    const unsigned wid = xcd_contiguous_id(blockIdx.x, gridDim.x);
    const int m0 = (int)(wid % p.mtiles) * 64, qt = (int)((wid / p.mtiles) % p.qtiles), seg = (int)(wid / (p.mtiles * p.qtiles));
    const int q0 = qt * NT, t0 = qt * 16;                        // first output position / frame of the tile
    const int Lin = p.Tin * V, Qout = p.Tout * V;
    const float *yseg = p.y + (int64_t)seg * p.C * Lin;

    f32x4 acc[NB];
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) acc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};

    // A-operand (activation) base of a column block: column c = 16 cb + i of the tile is output frame t0 + c / V; tap r reads
    // the staged row at  c + (S - 1) * (c / V) * V + r * V  (stride 2: the frames in between are staged too)
    int abase[S == 1 ? 1 : NB];
    if (S == 1) {
        abase[0] = kq * ROW + l15;
    } else {
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) {
            const int c = 16 * cb + l15;
            abase[cb] = kq * ROW + c + (c / V) * V;
        }
    }
    const float *wl_lane = Wl + kq * LDW + wave * 16 + l15;
    auto taps = [&](int r0, int r1) {
#pragma unroll
        for (int r = r0; r < r1; ++r) {
            if (r > r0) __builtin_amdgcn_sched_barrier(0);        // bounds the fragment read-ahead (registers) to one tap
#pragma unroll
            for (int s = 0; s < KCH / 4; ++s) {
                const float wf = wl_lane[(r * KCH + 4 * s) * LDW];
#pragma unroll
                for (int cb = 0; cb < NB; ++cb) {
                    const float a = S == 1 ? Bl[abase[0] + 4 * s * ROW + r * V + 16 * cb] : Bl[abase[S == 1 ? 0 : cb] + 4 * s * ROW + r * V];
                    acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wf, acc[cb], 0, 0, 0);
                }
            }
        }
    };
    {   // ---- phase 1: the 9-tap conv
        typename std::conditional<KCH == 8, W16x9x8<LDW>, W16<9, KCH, LDW>>::type ws;
        Clip16<KCH, G::NQ, ROW> rs;
        const int wstart = (S * t0 - p.pad) * V;                 // first staged input position (negative in the first tile)
        const bool edge = wstart < 0 || wstart + G::SPAN > Lin;  // uniform
        ws.setup(p.Cpad, p.Mpad, tid);
        rs.setup(wstart, Lin, tid);
        const float *wbase = p.w + m0;
        ws.issue(wbase);
        rs.issue(yseg, edge);
        for (int c0 = 0; c0 < p.Cpad; c0 += KCH) {
            __syncthreads();
            ws.commit(Wl);
            rs.commit(Bl);
            __syncthreads();
            // next chunk's loads in three bursts between the tap segments; past the end the last real chunk is loaded again into
            // the (then dead) staging registers so that the K loop stays one basic block.  Channels >= C (the zero-weight
            // padding of the packed operand): the LAST real chunk's rows are re-read (finite; their weights are zero) -- C is a
            // multiple of 8 here (checked by the launcher)
            const int cn = min(c0 + KCH, p.Cpad - KCH);
            const float *wnext = wbase + (size_t)cn * p.Mpad, *rnext = yseg + (int64_t)min(cn, p.C - KCH) * Lin;
            ws.issue_one(0, wnext);
            ws.issue_one(1, wnext);
            rs.template issue_third<0>(rnext, edge);
            __builtin_amdgcn_s_setprio(1);
            taps(0, 3);
            __builtin_amdgcn_s_setprio(0);
            ws.issue_one(2, wnext);
            ws.issue_one(3, wnext);
            rs.template issue_third<1>(rnext, edge);
            __builtin_amdgcn_s_setprio(1);
            taps(3, 6);
            __builtin_amdgcn_s_setprio(0);
            ws.issue_one(4, wnext);
            rs.template issue_third<2>(rnext, edge);
            __builtin_amdgcn_s_setprio(1);
            taps(6, 9);
            __builtin_amdgcn_s_setprio(0);
        }
    }
    // ---- phase 2: 1x1 residual conv on the (strided) block input: one tap, the tile's S * 15 + 1 input frames from res_off on
    if (p.res_mode == CSK_RES_CONV) {
        constexpr int KR = 8, ROWR = G::ROWR;
        float *Wr = smem, *Br = smem + KR * LDW;
        W16<1, KR, LDW> ws;
        Clip16<KR, G::SPAN_R / 4, ROWR> rs;
        const int Lres = p.Tres * V;
        const int wstart = (S * t0 + p.res_off) * V;
        const bool edge = wstart + G::SPAN_R > Lres || (wstart & 3) != 0;
        const float *xseg = p.xres + (int64_t)seg * p.Cres * Lres;
        ws.setup(p.CresPad, p.Mpad, tid);
        rs.setup(wstart & ~3, Lres, tid);                        // (a start that is not a multiple of 4 shifts the LDS row: see sub)
        const int sub = wstart & 3;
        const float *wbase = p.wres + m0;
        const float *wr_lane = Wr + kq * LDW + wave * 16 + l15;
        ws.issue(wbase);
        rs.issue(xseg, edge);
        for (int c0 = 0; c0 < p.CresPad; c0 += KR) {
            __syncthreads();
            ws.commit(Wr);
            rs.commit(Br);
            __syncthreads();
            const int cn = min(c0 + KR, p.CresPad - KR);
            ws.issue(wbase + (size_t)cn * p.Mpad);
            rs.issue(xseg + (int64_t)min(cn, max(p.Cres - KR, 0)) * Lres, edge);
#pragma unroll
            for (int s = 0; s < KR / 4; ++s) {
                const float wf = wr_lane[4 * s * LDW];
#pragma unroll
                for (int cb = 0; cb < NB; ++cb) {
                    const int c = 16 * cb + l15;
                    const float a = Br[(kq + 4 * s) * ROWR + sub + c + (S - 1) * (c / V) * V];
                    acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wf, acc[cb], 0, 0, 0);
                }
            }
        }
    }
    // ---- epilogue: + bias (+ identity residual at frame t' + res_off), ReLU; rows are Tout * V positions long
    const unsigned oslot[1] = {0u};
    const unsigned xslot[1] = {(unsigned)(p.res_off * V * 4)};
    const int nval = min(NT, Qout - q0);
    epilogue16<NB, 1, NT, true>(acc, p.bias, p.Cout, m0 + wave * 16 + l15, kq, p.res_mode == CSK_RES_IDENTITY, p.relu != 0,
                                p.xres + (int64_t)seg * p.Cres * p.Tres * V, p.out + (int64_t)seg * p.Cout * Qout, xslot, oslot,
                                (int64_t)p.Tres * V, Qout, q0, nval, nval);
}

// K order of the graph conv for CIN real input channels (csrc/gcn.hip, gcn_entry in tile16.h): entry (channel pair s, subset r,
// channel 2 s + h) ascending, padding channels left out.  agg_row: (channel, subset) -> position in that order; agg_sub /
// agg_chan: position -> subset / channel.
template <int CIN>
__host__ __device__ constexpr int agg_row(int ci, int r) {
    int n = 0;
    for (int e = 0; e < gcn_entry(ci, r, 4); ++e) {              // entries in front of it whose channel is real
        const int h = e & 1, sr = e >> 1, s = sr / 4;
        n += (2 * s + h) < CIN ? 1 : 0;
    }
    return n;
}
template <int CIN>
__host__ __device__ constexpr int agg_entry(int j) {            // j-th real entry
    int n = 0;
    for (int e = 0; e < 64; ++e) {
        const int h = e & 1, s = (e >> 1) / 4;
        if ((2 * s + h) < CIN) {
            if (n == j) return e;
            ++n;
        }
    }
    return 0;
}
// ... as tables of a constexpr object (evaluated by the compiler, whatever the optimiser makes of the loops around their use)
template <int CIN>
struct AggTab {
    int sub[4 * CIN], chan[4 * CIN], row[CIN][4];
    constexpr AggTab() : sub{}, chan{}, row{} {
        for (int j = 0; j < 4 * CIN; ++j) {
            const int e = agg_entry<CIN>(j);
            sub[j] = (e >> 1) % 4;
            chan[j] = 2 * ((e >> 1) / 4) + (e & 1);
        }
        for (int ci = 0; ci < CIN; ++ci)
            for (int r = 0; r < 4; ++r) row[ci][r] = agg_row<CIN>(ci, r);
    }
};

// ------------------------------------------------------------------------------------------------
// Block with a FEW input channels (layer 1 of the stacks: C_in = 3) as ONE launch: graph conv (3 adjacency subsets + 1x1 conv
// gcn_residual, BN, ReLU: models/base.py:230-270) formed on the fly inside the temporal conv's tile (models/base.py:376-387,
// no block residual: st_gcn.py:30).  The two-launch form writes and re-reads y = 64 channels (2 GB at batch 256) for a graph
// conv of 12 multiply-adds per output; here the tile's input window (C_in x 24 frames) is staged once, its 4 C_in aggregated
// rows (3 subsets + the input itself) are kept in LDS, and every 8-channel chunk of y is a small MFMA product of them (K = 4 C_in)
// written right into the operand rows the temporal conv's MFMAs read -- the temporal halo is recomputed (24 of 16 frames): 3 MFMAs
// per 16 positions and chunk beside the 450 of the temporal conv.  y is BIT FOR BIT the graph-conv kernel's: the same fmaf chain in its K order ((channel pair, subset, channel),
// csrc/gcn.hip), bias added last, ReLU, and ZERO in the frames of the temporal padding; the temporal conv is tcn_stage16_kernel's.
struct Fused1Params {
    TcnParams t;                   // the temporal conv (y unused); C = C_out of the graph conv
    const float *x, *gw, *gbias;   // block input (n_seg, Cin, Tin, V); packed graph-conv weights [4][CinPad][GMpad], bias
    const int *ell_src;
    const float *ell_val;
    int ell_cnt[3], ell_w, Cin, CinPad, GMpad;
};

template <int V, int CIN>
__global__ __launch_bounds__(NTHREADS, 2) void block1_fused16_kernel(const Fused1Params f) {
    typedef C16<V, 1, 8> G;
    constexpr int NB = G::NB, NT = G::NT, KCH = 8, LDW = G::LDW, ROW = G::ROW, SPAN = G::SPAN, NCOL = (SPAN + NTHREADS - 1) / NTHREADS;
    constexpr int NA = 4 * CIN;                                  // aggregated rows, in the graph conv's K order
    constexpr AggTab<CIN> TAB{};
    const TcnParams &p = f.t;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Wl = smem, *Bl = smem + G::WSZ, *Ag = Bl + KCH * ROW, *Xw = Bl;   // Xw [CIN][SPAN]: prologue only, in the operand rows' place
    static_assert(CIN * SPAN <= KCH * ROW, "the input window fits the operand rows");
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kq = lane >> 4;
    const unsigned wid = xcd_contiguous_id(blockIdx.x, gridDim.x);
    const int m0 = (int)(wid % p.mtiles) * 64, qt = (int)((wid / p.mtiles) % p.qtiles), seg = (int)(wid / (p.mtiles * p.qtiles));
    const int q0 = qt * NT, t0 = qt * 16;
    const int Lin = p.Tin * V, Qout = p.Tout * V;
    const int wstart = (t0 - p.pad) * V;                         // first staged position (a frame boundary; negative in the first tile)
    const float *xseg = f.x + (int64_t)seg * f.Cin * Lin;

    f32x4 acc[NB];
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) acc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    W16x9x8<LDW> ws;
    ws.setup(p.Cpad, p.Mpad, tid);
    const float *wbase = p.w + m0;
    ws.issue(wbase);
    // ---- prologue: the input window, zero outside the sequence; then the aggregated rows of this thread's columns
    for (int e = tid; e < CIN * SPAN; e += NTHREADS) {
        const int ci = e / SPAN, i = e - ci * SPAN, pos = wstart + i;
        Xw[e] = (pos >= 0 && pos < Lin) ? xseg[(int64_t)ci * Lin + pos] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int n = 0; n < NCOL; ++n) {
        const int i = min(n * NTHREADS + tid, SPAN - 1);         // (threads past the window redo its last column: the same values)
        const int lf = i / V, w = i - lf * V;
        float ev[6];
        int es[6];
#pragma unroll
        for (int e = 0; e < 6; ++e) {
            const int r = e < 2 ? e : 2, k = e < 2 ? 0 : e - 2;
            const bool have = k < f.ell_cnt[r];
            const int idx = (r * V + w) * f.ell_w + min(k, f.ell_w - 1);
            es[e] = lf * V + (have ? f.ell_src[idx] : 0);
            ev[e] = have ? f.ell_val[idx] : 0.f;
        }
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) {
            const float *xr = Xw + ci * SPAN;
            const float b0 = ev[0] * xr[es[0]];
            const float b1 = ev[1] * xr[es[1]];
            float s2 = ev[2] * xr[es[2]];
            s2 = fmaf(ev[3], xr[es[3]], s2);
            s2 = fmaf(ev[4], xr[es[4]], s2);
            s2 = fmaf(ev[5], xr[es[5]], s2);
            // (row = position of (channel, subset) in the graph conv's K order)
            Ag[TAB.row[ci][0] * SPAN + i] = b0;
            Ag[TAB.row[ci][1] * SPAN + i] = b1;
            Ag[TAB.row[ci][2] * SPAN + i] = s2;
            Ag[TAB.row[ci][3] * SPAN + i] = xr[i];
        }
    }
    // graph-conv weight of aggregated row 4 s + kq (the lane's k index), as an offset into the packed [4][CinPad][GMpad] operand
    constexpr int NCB = (SPAN + 15) / 16;
    static_assert(16 * NCB <= ROW, "the last column block stays inside an operand row");
    int woff[NA / 4];
#pragma unroll
    for (int s = 0; s < NA / 4; ++s) {
        int sub = 0, chan = 0;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            sub = (j == 4 * s + kq) ? TAB.sub[j] : sub;
            chan = (j == 4 * s + kq) ? TAB.chan[j] : chan;
        }
        woff[s] = (sub * f.CinPad + chan) * f.GMpad;
    }
    const float *wl_lane = Wl + kq * LDW + wave * 16 + l15;
    const int abase = kq * ROW + l15;
    auto taps = [&](int r0, int r1) {
#pragma unroll
        for (int r = r0; r < r1; ++r) {
            if (r > r0) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < KCH / 4; ++s) {
                const float wf = wl_lane[(r * KCH + 4 * s) * LDW];
#pragma unroll
                for (int cb = 0; cb < NB; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(Bl[abase + 4 * s * ROW + r * V + 16 * cb], wf, acc[cb], 0, 0, 0);
            }
        }
    };
    // graph-conv weights / bias of the lane's channel c0 + (l15 & 7), loaded one chunk ahead
    float wg[NA / 4], bv;
    auto load_wg = [&](int c0) {
        const int cw = min(c0 + (l15 & 7), p.C - 1);             // (channels past C: zero temporal weights)
#pragma unroll
        for (int s = 0; s < NA / 4; ++s) wg[s] = f.gw[woff[s] + cw];
        bv = f.gbias[cw];
    };
    float wgn[NA / 4], bvn;
    load_wg(0);
    // ---- K loop over 8-channel chunks of y
    for (int c0 = 0; c0 < p.Cpad; c0 += KCH) {
        __syncthreads();                                         // the previous chunk's operand reads are done (first: Ag is complete)
        ws.commit(Wl);
        {   // y rows of the chunk: D[16 positions][16 channels] = agg[16 pos][NA] x Wg[NA][channels] by 16x16x4 MFMAs -- the fmaf
            // chain of the graph-conv kernel in its K order (an MFMA sums its k values in ascending order); the lanes of channels
            // 8-15 repeat 0-7 and are not stored.  A lane holds 4 consecutive positions of channel c0 + (l15 & 7).
            // (the chunk's graph-conv weights and bias were loaded under the previous chunk's MFMAs; three column blocks at a time:
            // three independent accumulation chains)
            constexpr int CBU = 3, WV = NTHREADS / 64;
            for (int cb0 = wave; cb0 < NCB; cb0 += CBU * WV) {
                f32x4 d[CBU];
                float av[CBU][NA / 4];
#pragma unroll
                for (int u = 0; u < CBU; ++u) {
                    d[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                    const int ci = min(16 * (cb0 + u * WV) + l15, SPAN - 1);
#pragma unroll
                    for (int s = 0; s < NA / 4; ++s) av[u][s] = Ag[(4 * s + kq) * SPAN + ci];
                }
#pragma unroll
                for (int s = 0; s < NA / 4; ++s)
#pragma unroll
                    for (int u = 0; u < CBU; ++u) d[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][s], wg[s], d[u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < CBU; ++u) {
                    const int cb = cb0 + u * WV;
                    if (l15 < 8 && cb < NCB) {
                        const int i0 = 16 * cb + 4 * kq;
                        f32x4 o;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int pos = wstart + i0 + q;
                            o[q] = (pos >= 0 && pos < Lin) ? relu_nan(d[u][q] + bv + 0.f) : 0.f;   // zero in the temporal padding
                        }
                        *reinterpret_cast<f32x4 *>(Bl + l15 * ROW + i0) = o;
                    }
                }
            }
        }
        __syncthreads();
        const int cn = min(c0 + KCH, p.Cpad - KCH);
        const float *wnext = wbase + (size_t)cn * p.Mpad;
        {
            const int cw = min(cn + (l15 & 7), p.C - 1);
#pragma unroll
            for (int s = 0; s < NA / 4; ++s) wgn[s] = f.gw[woff[s] + cw];
            bvn = f.gbias[cw];
        }
        ws.issue_one(0, wnext);
        ws.issue_one(1, wnext);
        __builtin_amdgcn_s_setprio(1);
        taps(0, 3);
        __builtin_amdgcn_s_setprio(0);
        ws.issue_one(2, wnext);
        ws.issue_one(3, wnext);
        __builtin_amdgcn_s_setprio(1);
        taps(3, 6);
        __builtin_amdgcn_s_setprio(0);
        ws.issue_one(4, wnext);
        __builtin_amdgcn_s_setprio(1);
        taps(6, 9);
        __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int s = 0; s < NA / 4; ++s) wg[s] = wgn[s];
        bv = bvn;
    }
    const unsigned oslot[1] = {0u}, xslot[1] = {0u};
    const int nval = min(NT, Qout - q0);
    epilogue16<NB, 1, NT, true>(acc, p.bias, p.Cout, m0 + wave * 16 + l15, kq, false, p.relu != 0, p.out, p.out + (int64_t)seg * p.Cout * Qout,
                                xslot, oslot, (int64_t)Qout, Qout, q0, nval, nval);
}

template <int V, int S, int KCH_ = (S == 1 ? 8 : 4)>
int launch_stage16(TcnParams p, int n_seg, hipStream_t s) {
    typedef C16<V, S, KCH_> G;
    const int Q = p.Tout * V;
    p.qtiles = (unsigned)((Q + G::NT - 1) / G::NT); p.mtiles = (unsigned)(p.Mpad / 64);
    const int64_t grid = (int64_t)p.qtiles * p.mtiles * n_seg;
    if (grid >= (1ll << 31)) CSK_FAIL("tcn_stage: grid too large");
    void (*kern)(TcnParams) = tcn_stage16_kernel<V, S, KCH_>;
    const size_t lds = (size_t)G::LDS_FLOATS * sizeof(float);
    if (const int e = csk_ensure_lds((const void *)kern, lds)) return e;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NTHREADS), lds, s, p);
    return (int)hipGetLastError();
}

}  // namespace

// Which launches take this family is a function of the LAYER only (never of the batch): measured in-process against the 32x32x2
// stage kernel at batch 256 (tools/ab_probe.py CSK_TCN16=1, profiles/HISTORY.md round 6; C = channels of the temporal conv) it
// wins where the K loop is short -- C = 64: 2.282 vs 2.380 ms (-4.1 %); C = 128, stride 2: 4.853 vs 4.923 (-1.4 %) -- ties at
// C = 128, stride 1 and loses 1.8 % / 0.5 % at C = 256 (stride 2 / 1): both families end at 0.80-0.83 of the fp32 MFMA peak on long
// K loops (two waves per SIMD, each bound by the LDS round trip of its operand reads while its partner is outside its MFMA
// segment).  Taken for C <= 64 and for stride 2 at C <= 128.  V = 18 (Kinetics): every layer -- 16 frames are 288 = 18 x 16 columns
// exactly where the 32-wide tiles hold 180 of 192 (A-GCN clip forward, batch 64, same process: 16.74 -> 16.53 ms; C = 128 stride 1
// 0.893 -> 0.837 ms, C = 256 stride 1 1.756 -> 1.672, C = 256 stride 2 1.897 -> 1.879).
// CSK_TCN16 under CSK_DIAG=1: 1 = never, 2 = every shape the kernel supports (A/B runs, parity tests of the wide layers).
int csk_launch_tcn_stage16(TcnParams p, int n_seg, void *stream) {
    const int mode = csk_diag_int("CSK_TCN16");
    if (mode == 1) return -2;
    if (mode != 2 && p.V != 18 && p.C > (p.stride == 2 ? 128 : 64)) return -2;
    if (p.K != 9 || p.ksplit != 1 || (p.V != 25 && p.V != 18) || p.stride < 1 || p.stride > 2) return -2;
    if ((p.pad * p.V) & 3) return -2;                            // the staged window starts on a 16-byte LDS boundary
    if (p.C % 8 != 0 || p.C < 8) return -2;                      // whole chunks of real channels (the stack's convs: 64 / 128 / 256)
    if (p.res_mode == CSK_RES_CONV && (p.Cres % 8 != 0 || p.Cres < 8)) return -2;
    if (p.res_mode == CSK_RES_IDENTITY && p.stride != 1) return -2;
    if (p.res_mode == CSK_RES_CONV && ((p.res_off * p.V) & 3)) return -2;
    // 32-bit byte offsets inside a segment
    if ((int64_t)p.C * p.Tin * p.V * 4 >= (1ll << 31) || (int64_t)p.Cout * p.Tout * p.V * 4 >= (1ll << 31) ||
        (int64_t)p.Cres * p.Tres * p.V * 4 >= (1ll << 31))
        return -2;
    if (p.Tin * p.V < 8 || (p.res_mode != CSK_RES_NONE && p.Tres * p.V < 8)) return -2;
    hipStream_t s = (hipStream_t)stream;
    if (p.V == 25) return p.stride == 1 ? launch_stage16<25, 1>(p, n_seg, s) : launch_stage16<25, 2>(p, n_seg, s);
    return p.stride == 1 ? launch_stage16<18, 1>(p, n_seg, s) : launch_stage16<18, 2>(p, n_seg, s);
}

template <int V, int CIN>
static int launch_fused1(Fused1Params f, int n_seg, hipStream_t s) {
    typedef C16<V, 1, 8> G;
    TcnParams &p = f.t;
    const int Q = p.Tout * V;
    p.qtiles = (unsigned)((Q + G::NT - 1) / G::NT); p.mtiles = (unsigned)(p.Mpad / 64);
    const int64_t grid = (int64_t)p.qtiles * p.mtiles * n_seg;
    if (grid >= (1ll << 31)) CSK_FAIL("block1_fused: grid too large");
    void (*kern)(Fused1Params) = block1_fused16_kernel<V, CIN>;
    const size_t lds = (size_t)(G::WSZ + 8 * G::ROW + 4 * CIN * G::SPAN) * sizeof(float);
    if (const int e = csk_ensure_lds((const void *)kern, lds)) return e;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NTHREADS), lds, s, f);
    return (int)hipGetLastError();
}

// C ABI: see include/cskel.h (csk_block_few_channels_f32)
extern "C" int csk_block_few_channels_f32(const float *x, const float *gcn_w, const float *gcn_bias, const int32_t *ell_src,
                                          const float *ell_val, const int32_t *ell_cnt, int ell_w, const float *tcn_w,
                                          const float *tcn_bias, float *out, int n_seg, int c_in, int c_mid, int c_out, int t_in,
                                          int V, int pad, void *stream) {
    if (!x || !gcn_w || !gcn_bias || !ell_src || !ell_val || !ell_cnt || !tcn_w || !tcn_bias || !out) CSK_FAIL("block_few_channels: null pointer");
    if (n_seg <= 0 || t_in <= 0 || c_out <= 0) CSK_FAIL("block_few_channels: bad dims");
    if (c_in < 1 || c_in > 4) CSK_FAIL("block_few_channels: built for 1..4 input channels (layer 1 of the stacks)");
    if (V != 25 && V != 18) CSK_FAIL("block_few_channels: built for V = 25 / 18");
    if (pad != 4 || c_mid < 8 || c_mid % 8) CSK_FAIL("block_few_channels: 9-tap temporal conv with padding 4, graph-conv channels a multiple of 8");
    if (ell_cnt[0] > 1 || ell_cnt[1] > 1 || ell_cnt[2] > 4 || ell_w < 1) CSK_FAIL("block_few_channels: skeleton-sparse adjacency (<= 1 / 1 / 4 entries per column)");
    if ((int64_t)c_mid * t_in * V * 4 >= (1ll << 31) || (int64_t)c_out * t_in * V * 4 >= (1ll << 31)) CSK_FAIL("block_few_channels: segment too large");
    Fused1Params f{};
    TcnParams &p = f.t;
    p.w = tcn_w; p.bias = tcn_bias; p.out = out;
    p.C = c_mid; p.Cpad = round_up(c_mid, CSK_CPAD); p.Cout = c_out; p.Mpad = round_up(c_out, CSK_MT);
    p.Tin = t_in; p.Tout = t_in; p.V = V; p.K = 9; p.stride = 1; p.pad = pad; p.relu = 1; p.res_mode = CSK_RES_NONE; p.ksplit = 1;
    f.x = x; f.gw = gcn_w; f.gbias = gcn_bias; f.ell_src = ell_src; f.ell_val = ell_val;
    for (int k = 0; k < 3; ++k) f.ell_cnt[k] = ell_cnt[k];
    f.ell_w = ell_w; f.Cin = c_in; f.CinPad = round_up(c_in, CSK_CPAD); f.GMpad = round_up(c_mid, CSK_MT);
    hipStream_t s = (hipStream_t)stream;
#define CSK_F1(V_) (c_in == 1 ? launch_fused1<V_, 1>(f, n_seg, s) : c_in == 2 ? launch_fused1<V_, 2>(f, n_seg, s) : c_in == 3 ? launch_fused1<V_, 3>(f, n_seg, s) : launch_fused1<V_, 4>(f, n_seg, s))
    return V == 25 ? CSK_F1(25) : CSK_F1(18);
#undef CSK_F1
}
