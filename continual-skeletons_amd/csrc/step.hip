// step.hip -- gfx950 kernels for the CoST-GCN continual (frame-by-frame) path.
//
// State lives in a persistent HBM slab in CHANNEL-MAJOR layout: one frame of activations for all streams is
// a (C, P) matrix, P = n_streams * M * V positions (padded to a multiple of 4), position = (skeleton, joint)
// with the joint innermost.  Per block (depths: CSK_CO_YRING / CSK_CO_HIST = 16 / 16, include/cskel.h):
//     y ring   [16][C_out][P]   post-GCN frames  (the (k-1) = 8-frame window of co.Conv2d + up to 8 new frames of a
//                                launch cycle)
//     out ring [16][C_out][P]   block outputs    (doubles as the next block's input / residual history, co.Delay(4),
//                                plus the frames of a cycle in flight)
// A cycle of a block = gcn_stage on the new frames (gcn.hip, frames == skeletons) writing ring slots s % 16 ..., then
// tcn_step below for the emitting steps: the 9 taps of the temporal conv are 9 consecutive ring slots at the SAME
// positions, so the GEMM is  D[co, p] = sum_r sum_c W[r][c][co] * ring[(head - 8 + r) % slots][c][p].
// Zero-initialised slots reproduce the zero left-padding of the clip conv (models/base.py:307-334 semantics).
// (A bare CoTemporalConvolution uses the same kernel on a ring of exactly k slots.)
#include <type_traits>

#include "mfma_core.h"
#include "step_params.h"

// ------------------------------------------------------------------------------------------------
// ring staging: NS ring slots x KC channel rows x NP positions -> Bl [slot][KC][NP]
// ------------------------------------------------------------------------------------------------
// A workgroup tile of NT = 16384/MT columns is E emissions x NP = NT/E positions: the E emissions of a launch cycle
// at the same positions share all but (E-1)*head_step of their 9 ring slots, so one staged window of
//     S = K-1 + (E-1)*head_step + 1   slots
// serves E*K taps (E = 4: 12 slots instead of 36 -- the step analogue of the clip kernel's temporal halo), and a tap
// of emission j is the LDS row block  j*head_step + r.
// Staging sweep u of a thread covers row u*RPU + tid/N4 of the [slot][kk] row space (RPU = 256/N4 rows per sweep).
// A wave covers whole rows of ONE slot in every sweep (N4 <= 64, rows per wave divide KC), so the slot of a sweep is
// wave-uniform: slot bases stay on the scalar unit.  Rows past the window (sweep padding, K < 9) re-stage its last
// slot into LDS rows that are never read.
template <int NP, int NS>
struct RingStage {
    static constexpr int N4 = NP / 4;
    static constexpr int RPU = NTHREADS / N4;                       // rows per sweep: 4 (NP=256) .. 16 (NP=64)
    // window() makes a sweep's ring-slot offset wave-uniform with readfirstlane: legal only if a wave's rows of a sweep
    // (64 / N4 of them) never straddle two slots, i.e. they divide the KC rows of a slot
    static_assert(NP <= 256 && N4 <= 64 && KC % (64 / N4) == 0, "a wave's rows of a sweep must lie inside one ring slot");
    static constexpr int NB = (NS * KC + RPU - 1) / RPU;            // f32x4 per thread per chunk
    static constexpr int LDS_FLOATS = NB * RPU * NP;
    unsigned poff;                                                  // clamped position offset inside the tile
    int krow;                                                       // tid / N4: row inside a sweep
    int64_t soff[NB];                                               // element offset of this wave's ring slot per sweep (uniform)
    f32x4 v[NB];
    __device__ __forceinline__ void setup(int p0, int64_t P, int tid) {
        krow = tid / N4;
        poff = (unsigned)min((tid % N4) * 4, (int)(P - 4 - p0));    // last legal f32x4 start relative to p0
    }
    // window slot w lives in ring slot (first + w*step) mod slots of Cs channel rows each (computed once per phase)
    __device__ __forceinline__ void window(int first, int step, int slots, int nwin, int Cs, int64_t P) {
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int w = min((u * RPU + krow) / KC, nwin - 1);
            const int64_t o = (int64_t)((first + w * step) % slots) * Cs * P;
            soff[u] = ((int64_t)__builtin_amdgcn_readfirstlane((int)(o >> 32)) << 32) |
                      (unsigned)__builtin_amdgcn_readfirstlane((int)o);
        }
    }
    // base = ring + p0 (uniform); rows c0 + kk, channels >= C read as zero
    template <int U0, int U1>
    __device__ __forceinline__ void issue_range(const float *__restrict__ base, int C, int64_t P, int c0) {
#pragma unroll
        for (int u = U0; u < U1; ++u) {
            const int c = c0 + (u * RPU + krow) % KC;
            const f32x4 x = *reinterpret_cast<const f32x4 *>(base + soff[u] + (int64_t)min(c, C - 1) * P + poff);
            v[u] = x * (c < C ? 1.f : 0.f);
        }
    }
    // NU = sweeps actually needed (compile-time: ceil(window_slots*KC / RPU))
    template <int NU = NB>
    __device__ __forceinline__ void issue(const float *__restrict__ base, int C, int64_t P, int c0) {
        issue_range<0, (NU < NB ? NU : NB)>(base, C, P, c0);
    }
    // one third of the chunk's sweeps (G is a literal so that register indices are static)
    template <int G>
    __device__ __forceinline__ void issue_third(const float *__restrict__ base, int C, int64_t P, int c0) {
        issue_range<G * NB / 3, (G + 1) * NB / 3>(base, C, P, c0);
    }
    template <int NU = NB>
    __device__ __forceinline__ void commit(float *__restrict__ Bl) const {
#pragma unroll
        for (int u = 0; u < (NU < NB ? NU : NB); ++u) *reinterpret_cast<f32x4 *>(Bl + (u * RPU + krow) * NP + poff) = v[u];
    }
};


// E = emissions per workgroup tile (folded into the tile's column axis, see RingStage), HS = head_step of the launch
// (1, or 2 for a stride-2 block; only meaningful for E > 1).  SPLIT = false is the throughput kernel; the split-K form
// is a separate instantiation (E = 1) so that its extra index arithmetic costs the default path nothing.
template <int MT, int E, int HS, bool SPLIT, bool K9 = true, bool HALVES = false>
__device__ __forceinline__ void tcn_step_tile(const StepParams &p, const int bx, const int by, const int bz, float *smem) {
    constexpr int NT = 16384 / MT;
    constexpr int NP = NT / E;                      // positions per tile
    constexpr int NS = 8 + (E - 1) * HS + 1;        // window slots staged per chunk (K = 9)
    constexpr int WM = MT / 64;
    static_assert(NP >= 64, "a wave's 64 columns must belong to one emission");
    typedef RingStage<NP, NS> Stage;
    constexpr int NU_RES = (E * KC + Stage::RPU - 1) / Stage::RPU;      // sweeps of the residual-conv phase (E slots)
    float *Wl = smem;                       // [K][KC][MT]
    float *Bl = smem + 9 * KC * MT;         // [NS (+ sweep padding)][KC][NP]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    const int l31 = lane & 31, kh = lane >> 5;
    const int m0 = by * MT, p0 = bx * NP;
    const int64_t P = p.P;
    // emission group of this workgroup (bz = blockIdx.z): emissions j0 .. j0+E-1 of the launch; emission j has its newest
    // frame in slot head + j*head_step, its residual frame in xres slot xres_slot0 + j*xres_step and goes to out slot
    // out_slot0 + j (all modulo their ring depths)
    constexpr bool split = SPLIT;
    const int ks = split ? bz % p.ksplit : 0;
    const int j0 = (split ? bz / p.ksplit : bz) * E;
    const int jw = j0 + (wn * 64) / NP;                       // emission of this wave's 64 columns (wave-uniform)
    const int cb = ks * p.cper;                               // first channel of this split
    const int Cl = split ? min(p.C - cb, p.cper) : p.C;       // its channels (<= 0: padding-only split, sums stay zero)
    const int CpadL = split ? min(p.Cpad - cb, p.cper) : p.Cpad;
    const int nwin = p.K + (E - 1) * HS;                      // window slots actually used
    int first = (p.head + j0 * p.head_step - (p.K - 1)) % p.slots;     // ring slot of window slot 0
    if (first < 0) first += p.slots;
    const float *xres = p.xres + (int64_t)((p.xres_slot0 + jw * p.xres_step) % p.xres_slots) * p.Cres * P;
    float *out = split ? p.part + (int64_t)(jw * p.ksplit + ks) * p.Cout * P
                       : p.out + (int64_t)((p.out_slot0 + jw) % p.out_slots) * p.Cout * P;

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[a][b][g] = 0.f;

    const int offA = wm * 64 + l31;
    // B operand of (tap r, channel kk, this lane's column): Bl[((jl*HS + r)*KC + kk)*NP + column inside the emission]
    const int jl = (wn * 64) / NP, cw = (wn * 64) % NP;
    const int off0 = jl * HS * KC * NP + cw + l31, off1 = off0 + 32;
    WStage<MT> ws;
    Stage rs;
    // ---- phase 1: temporal conv over the ring window
    {
        const float *wbase = p.w + m0 + (size_t)cb * p.Mpad;
        const float *rbase = p.ring + (int64_t)cb * P + p0;
        // 9-tap chunks: the compact weight staging of mfma_core.h (one offset register instead of 2 x WB)
        using WS9 = typename std::conditional<MT == 128, WStage9x128, WStage9x64>::type;
        typename std::conditional<K9, WS9, WStage<MT> &>::type ws1 = [&]() -> decltype(auto) {
            if constexpr (K9) { WS9 w9; w9.setup(p.Cpad, p.Mpad, tid); return w9; }
            else { ws.setup(p.K, p.Cpad, p.Mpad, tid); return (ws); }
        }();
        rs.setup(p0, P, tid);
        ws1.issue(wbase);
        rs.window(first, 1, p.slots, nwin, p.C, P);
        rs.issue(rbase, Cl, P, 0);
        const int t1 = (p.K + 2) / 3, t2 = min(p.K, 2 * t1);
        constexpr bool k9 = K9;                          // 9 taps: straight-line 3-tap MFMA segments (mfma_taps_ct; spill-free at 155-236 VGPRs, kernel_resources.txt)
        int c0 = 0;
        for (; c0 + KC < CpadL; c0 += KC) {
            __syncthreads();
            ws1.commit(Wl);
            rs.commit(Bl);
            __syncthreads();
            // next chunk's loads in three bursts between three tap segments (see mfma_taps in mfma_core.h)
            const float *wnext = wbase + (size_t)(c0 + KC) * p.Mpad;
#pragma unroll
            for (int j = 0; j < 3; ++j) ws1.issue_slot(j, wnext);
            rs.template issue_third<0>(rbase, Cl, P, c0 + KC);
            __builtin_amdgcn_s_setprio(1);
            if (k9) mfma_taps_ct<MT, 3>(Wl, Bl, 0, NP, KC * NP, offA, off0, off1, kh, acc);
            else mfma_taps<MT>(Wl, Bl, 0, t1, NP, KC * NP, offA, off0, off1, kh, acc);
            __builtin_amdgcn_s_setprio(0);
#pragma unroll
            for (int j = 3; j < 6; ++j) ws1.issue_slot(j, wnext);
            rs.template issue_third<1>(rbase, Cl, P, c0 + KC);
            __builtin_amdgcn_s_setprio(1);
            if (k9) mfma_taps_ct<MT, 3>(Wl, Bl, 3, NP, KC * NP, offA, off0, off1, kh, acc);
            else if (t1 < t2) mfma_taps<MT>(Wl, Bl, t1, t2, NP, KC * NP, offA, off0, off1, kh, acc);
            __builtin_amdgcn_s_setprio(0);
#pragma unroll
            for (int j = 6; j < 9; ++j) ws1.issue_slot(j, wnext);
            rs.template issue_third<2>(rbase, Cl, P, c0 + KC);
            __builtin_amdgcn_s_setprio(1);
            if (k9) mfma_taps_ct<MT, 3>(Wl, Bl, 6, NP, KC * NP, offA, off0, off1, kh, acc);
            else if (t2 < p.K) mfma_taps<MT>(Wl, Bl, t2, p.K, NP, KC * NP, offA, off0, off1, kh, acc);
            __builtin_amdgcn_s_setprio(0);
        }
        __syncthreads();
        ws1.commit(Wl);
        rs.commit(Bl);
        __syncthreads();
        mfma_chunk<MT>(Wl, Bl, p.K, NP, KC * NP, offA, off0, off1, kh, acc);
    }
    // ---- phase 2: 1x1 residual conv on the delayed block input (CoTempConv k=1 + co.Delay, base.py:424-441): the
    // window is the E residual frames of the tile's emissions, one "tap" each
    if (p.res_mode == CSK_RES_CONV && ks == 0) {
        const float *wbase = p.wres + m0;
        const float *xbase = p.xres + p0;
        const int xfirst = (p.xres_slot0 + j0 * p.xres_step) % p.xres_slots;
        const int offr0 = jl * KC * NP + cw + l31, offr1 = offr0 + 32;
        ws.setup(1, p.CresPad, p.Mpad, tid);
        ws.issue(wbase);
        rs.window(xfirst, p.xres_step, p.xres_slots, E, p.Cres, P);
        rs.template issue<NU_RES>(xbase, p.Cres, P, 0);
        for (int c0 = 0; c0 < p.CresPad; c0 += KC) {
            __syncthreads();
            ws.commit(Wl);
            rs.template commit<NU_RES>(Bl);
            __syncthreads();
            if (c0 + KC < p.CresPad) {
                ws.issue(wbase + (size_t)(c0 + KC) * p.Mpad);
                rs.template issue<NU_RES>(xbase, p.Cres, P, c0 + KC);
            }
            mfma_chunk<MT>(Wl, Bl, 1, NP, KC * NP, offA, offr0, offr1, kh, acc);
        }
    }
    // ---- epilogue: + bias (+ identity residual), ReLU, stores.  Same scheme as tcn_stage_kernel: on full tiles the row
    // base pointers are wave-uniform (scalar unit) and every access carries one 32-bit lane byte offset;
    // v_permlane32_swap pairs the ni = 0 / 1 registers so that a store instruction writes one 256-B row segment.
    // (split-K: raw partial sums -- bias, identity residual and ReLU are applied by step_reduce_kernel)
    const bool ident = !split && p.res_mode == CSK_RES_IDENTITY;
    const bool relu = !split && p.relu;
    const int rbase = m0 + wm * 64;
    const bool full = p.fast_epi && m0 + MT <= p.Cout;
    const unsigned kh4 = 4u * (unsigned)kh;
    const int pw = p0 + cw;                                   // first position of this wave's columns
    // Operands are loaded and consumed one 32-row half (mi) at a time when HALVES (the 168-register instantiation: 64
    // accumulators + 2 x 48 operands would not fit), else both halves up front (all loads in flight together).
    float bv[2][16], rv[2][2][16];
    auto load_half = [&](int mi) {
        if (full) {
#pragma unroll
            for (int g = 0; g < 16; ++g)
                bv[mi][g] = split ? 0.f : ld_lane(p.bias + (rbase + mi * 32 + (g & 3) + 8 * (g >> 2)), kh4 * 4u);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const unsigned lo = 4u * (kh4 * (unsigned)P + (unsigned)min((int64_t)(pw + ni * 32 + l31), P - 1));
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const float *rrow = xres + (int64_t)(rbase + mi * 32 + (g & 3) + 8 * (g >> 2)) * P;
                    rv[ni][mi][g] = ident ? ld_lane(rrow, lo) : 0.f;
                }
            }
        } else {
#pragma unroll
            for (int g = 0; g < 16; ++g) bv[mi][g] = split ? 0.f : p.bias[rbase + mi * 32 + 4 * kh + (g & 3) + 8 * (g >> 2)];
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int64_t qc = min((int64_t)(pw + ni * 32 + l31), P - 1);
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const int co = rbase + mi * 32 + 4 * kh + (g & 3) + 8 * (g >> 2);
                    rv[ni][mi][g] = ident ? xres[(int64_t)min(co, p.Cout - 1) * P + qc] : 0.f;
                }
            }
        }
    };
    const int64_t qb = (int64_t)pw + lane;
    const bool qv = qb < P;
    auto finish_half = [&](int mi) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            float v0 = acc[mi][0][g] + bv[mi][g] + rv[0][mi][g];
            float v1 = acc[mi][1][g] + bv[mi][g] + rv[1][mi][g];
            if (relu) { v0 = relu_nan(v0); v1 = relu_nan(v1); }
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v0), __float_as_uint(v1), false, false);
            acc[mi][0][g] = __uint_as_float(sw[0]);            // row rbase + mi*32 + (g&3) + 8(g>>2), column qb
            acc[mi][1][g] = __uint_as_float(sw[1]);            // row + 4
        }
        if (full) {
            if (qv) {
                const unsigned qo = 4u * (unsigned)qb;
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    float *orow = out + (int64_t)(rbase + mi * 32 + (g & 3) + 8 * (g >> 2)) * P;
                    st_lane(orow, qo, acc[mi][0][g]);
                    st_lane(orow + 4 * P, qo, acc[mi][1][g]);
                }
            }
        } else {
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int row0 = rbase + mi * 32 + (g & 3) + 8 * (g >> 2);
                if (qv && row0 < p.Cout) out[(int64_t)row0 * P + qb] = acc[mi][0][g];
                if (qv && row0 + 4 < p.Cout) out[(int64_t)(row0 + 4) * P + qb] = acc[mi][1][g];
            }
        }
    };
    if (HALVES) {
        load_half(0);
        finish_half(0);
        __builtin_amdgcn_sched_barrier(0);
        load_half(1);
        finish_half(1);
    } else {
        load_half(0);
        load_half(1);
        finish_half(0);
        finish_half(1);
    }
}

// OCC = 3: the same kernel within 168 registers (three workgroups per CU), for launches whose workgroup count fits 768 resident
// slots in fewer rounds than 512 (CoAGCN at the Kinetics shape: 576 tiles per 64-channel block)
template <int MT, int E, int HS, bool SPLIT, bool K9, int OCC = 2>
__global__ __launch_bounds__(NTHREADS, OCC) void tcn_step_kernel(const StepParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // XCD-aware work-item order (mfma_core.h): every XCD walks a contiguous range of items; m-tile fastest, then emission
    // group, then position tile -- the m-tiles of a position tile read the SAME ring window (and emission groups at the
    // same positions all but a few slots of it); with a plain (x, y, z) grid they sat gridDim.x dispatches apart, by when
    // the window had left the 4 MB L2 (C = 256 launches fetched 966 MB for 525 MB of operands)
    const unsigned wid = xcd_contiguous_id(blockIdx.x, gridDim.x);
    tcn_step_tile<MT, E, HS, SPLIT, K9, OCC == 3>(p, (int)(wid / (p.gy * p.gz)), (int)(wid % p.gy), (int)((wid / p.gy) % p.gz), smem);
}

// ------------------------------------------------------------------------------------------------
// Fused continual block: GCN stage + TCN step of one CoSpatioTemporalBlock in ONE launch (models/base.py:412-446 is one
// forward_step).  For blocks with C_out <= 64 and stride 1 advancing a full stride cycle of E = 4 frames:
//   a workgroup owns NP = 64 positions of all four new frames.
//   phase G  wave w computes the post-GCN frame w at these positions (64 channels x 64 positions, K = R*C_in):
//            weights staged per 8-channel chunk for the whole workgroup, the wave's own input rows (all joints of the
//            <= 4 skeletons its positions touch) in a private LDS strip, aggregation formed on the fly from
//            register-resident adjacency entries exactly as in gcn_stage_sparse2_kernel (same summation order);
//            y = ReLU(acc + bias + gcn_residual) goes to the y-ring slot of frame w (it is state: later cycles need it).
//   hand-off the four new slots are read back by phase T through L2 (no HBM round trip): stores retired
//            (s_waitcnt), workgroup barrier, L1 invalidated (other workgroups of this CU may have pulled lines that
//            straddle the tile edge before they were written).
//   phase T  tcn_step_tile<64, 4, 1>: the 12-slot window (8 old + 4 new) serves the 4 x 9 taps; residual and output as
//            in the two-launch form.
// There is no temporal halo in step mode, so nothing is recomputed; what the fusion buys is one launch (and one
// partially filled tail round) per block instead of two, and the new frames' y read from L2 instead of HBM.  Measured
// (tools/online_pass.py [--no-fuse], 1024 streams): performance-neutral, 943-951 k vs 942-948 k frames/s with two
// stream shards, 817-820 k vs 819-822 k with one -- with two shards the GPU already runs two kernels 94 % of the time,
// so neither the saved launch nor the saved HBM read is on the critical path.  Kept as the default form of the blocks it
// covers because it is what models/base.py:412-446 is (one forward_step) and it halves their launch count.
// ------------------------------------------------------------------------------------------------
struct CoBlockParams {
    StepParams t;                        // phase T (ring = the y ring; head = slot of the first new frame)
    const float *xin;                    // block input ring [xin_slots][Cin][P]; new frame f in slot (xin_slot0 + f) % xin_slots
    const float *gw, *gbias;             // packed GCN operands (csk_gcn_stage_f32)
    const int32_t *ell_src;
    const float *ell_val;
    int ell_cnt[3], ell_w;
    int xin_slots, xin_slot0, Cin, CinPad, V, n_skel, ldbx, fast_epi_g;
    unsigned vmagic;
};

// (two workgroups per CU: 176 registers; cut to 168 for three per CU it spills 8-11 registers and measures the same)
template <bool CONVRES>
__global__ __launch_bounds__(NTHREADS, 2) void co_block_kernel(const CoBlockParams p) {
    constexpr int MT = 64, NP = 64, E = 4, KCG8 = 8;
    constexpr int R = CONVRES ? 4 : 3;
    constexpr int M4 = MT / 4;
    constexpr int WB = (R * KCG8 * M4 + NTHREADS - 1) / NTHREADS;   // 2
    constexpr int NJ = 2;                                           // 64-lane sweeps per private activation row (<= 128 positions)
    constexpr int NS = KCG8 / 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5;
    const int p0 = blockIdx.x * NP;
    const int64_t P = p.t.P;
    const int V = p.V, Q = p.n_skel * V;                            // valid positions (P is Q rounded up to 4)
    {   // ---------------- phase G: frame `wave`
        const int wsz = R * KCG8 * MT, xsz = E * KCG8 * p.ldbx, bufsz = wsz + xsz;   // chunk buffer: Wl, then 4 private strips
        const int qend = min(p0 + NP, Q);
        const int ta = div_magic(min(p0, Q - 1), p.vmagic), tb = div_magic(max(qend, 1) - 1, p.vmagic);
        const int span = (tb - ta + 1) * V;
        int eoff[2][6], ioff[2];
        float eval[2][6];
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int q = min(p0 + ni * 32 + l31, Q - 1);
            const int t = div_magic(q, p.vmagic);
            const int w = q - t * V, fb = max(t - ta, 0) * V;
            ioff[ni] = fb + w;
#pragma unroll
            for (int e = 0; e < 6; ++e) {
                const int r = e < 2 ? e : 2, k = e < 2 ? 0 : e - 2;
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
        const float *xf = p.xin + (int64_t)((p.xin_slot0 + wave) % p.xin_slots) * p.Cin * P;     // this wave's frame
        f32x4 wv[WB];
        unsigned wgo[WB], wlo[WB];
#pragma unroll
        for (int u = 0; u < WB; ++u) {
            const int e = min(u * NTHREADS + tid, R * KCG8 * M4 - 1);
            const int row = e / M4, m4 = e % M4;
            wgo[u] = (unsigned)(((row / KCG8) * p.CinPad + (row % KCG8)) * p.t.Mpad + m4 * 4);
            wlo[u] = (unsigned)(e * 4);
        }
        float bv[KCG8][NJ];
        unsigned bgo[NJ], blo[NJ];
#pragma unroll
        for (int u = 0; u < NJ; ++u) {
            const int j = min(u * 64 + lane, span - 1);
            bgo[u] = (unsigned)min(ta * V + j, Q - 1);
            blo[u] = (unsigned)j;
        }
        auto issue_all = [&](int c0) {
#pragma unroll
            for (int u = 0; u < WB; ++u) wv[u] = *reinterpret_cast<const f32x4 *>(p.gw + (size_t)c0 * p.t.Mpad + wgo[u]);
#pragma unroll
            for (int kk = 0; kk < KCG8; ++kk) {
                const int c = min(c0 + kk, p.Cin - 1);             // clamped: padding channels carry zero weights
#pragma unroll
                for (int u = 0; u < NJ; ++u) bv[kk][u] = (xf + (int64_t)c * P)[bgo[u]];
            }
        };
        auto commit = [&](float *buf) {
#pragma unroll
            for (int u = 0; u < WB; ++u) *reinterpret_cast<f32x4 *>(buf + wlo[u]) = wv[u];
            float *dst = buf + wsz + wave * (KCG8 * p.ldbx);
#pragma unroll
            for (int kk = 0; kk < KCG8; ++kk)
#pragma unroll
                for (int u = 0; u < NJ; ++u) dst[kk * p.ldbx + blo[u]] = bv[kk][u];
        };
        auto mfma_step = [&](const float *buf, int s) {
            const float *Wl = buf, *Bx = buf + wsz + wave * (KCG8 * p.ldbx);
            const int kk = 2 * s + kh;
            const float *bx = Bx + kk * p.ldbx;
            float b[R][2];
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                b[0][ni] = eval[ni][0] * bx[eoff[ni][0]];
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
                const float *wr = Wl + (r * KCG8 + kk) * MT + l31;
                const float a0 = wr[0], a1 = wr[32];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b[r][0], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b[r][1], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b[r][0], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b[r][1], acc[1][1], 0, 0, 0);
            }
        };
        const int nchunks = p.CinPad / KCG8;
        issue_all(0);
        commit(smem);
        if (nchunks > 1) issue_all(KCG8);
        __syncthreads();
        for (int c = 0; c + 1 < nchunks; ++c) {
            float *cur = smem + (c & 1) * bufsz, *oth = smem + ((c & 1) ^ 1) * bufsz;
            commit(oth);
            issue_all(min(c + 2, nchunks - 1) * KCG8);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int s = 0; s < NS; ++s) mfma_step(cur, s);
            __builtin_amdgcn_s_setprio(0);
            __syncthreads();
        }
        {
            const float *last = smem + ((nchunks - 1) & 1) * bufsz;
#pragma unroll
            for (int s = 0; s < NS; ++s) mfma_step(last, s);
        }
        // y = ReLU(acc + bias + identity gcn_residual) -> y ring slot of this wave's frame
        float *yf = p.t.ring == nullptr ? nullptr
                                        : const_cast<float *>(p.t.ring) + (int64_t)((p.t.head + wave) % p.t.slots) * p.t.C * P;
        const bool full = p.fast_epi_g && MT <= p.t.C;
        const unsigned kh4 = 4u * (unsigned)kh;
        const int qb = p0 + lane;
        const bool qv = qb < Q;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            float bb[16], rv[2][16];
            if (full) {
#pragma unroll
                for (int g = 0; g < 16; ++g) bb[g] = ld_lane(p.gbias + (mi * 32 + (g & 3) + 8 * (g >> 2)), kh4 * 4u);
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const unsigned lo = 4u * (kh4 * (unsigned)P + (unsigned)min(p0 + ni * 32 + l31, Q - 1));
#pragma unroll
                    for (int g = 0; g < 16; ++g)
                        rv[ni][g] = CONVRES ? 0.f : ld_lane(xf + (int64_t)(mi * 32 + (g & 3) + 8 * (g >> 2)) * P, lo);
                }
            } else {
#pragma unroll
                for (int g = 0; g < 16; ++g) bb[g] = p.gbias[mi * 32 + 4 * kh + (g & 3) + 8 * (g >> 2)];
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const int qc = min(p0 + ni * 32 + l31, Q - 1);
#pragma unroll
                    for (int g = 0; g < 16; ++g) {
                        const int co = mi * 32 + 4 * kh + (g & 3) + 8 * (g >> 2);
                        rv[ni][g] = CONVRES ? 0.f : xf[(int64_t)min(co, p.t.C - 1) * P + qc];
                    }
                }
            }
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const float v0 = relu_nan(acc[mi][0][g] + bb[g] + rv[0][g]);
                const float v1 = relu_nan(acc[mi][1][g] + bb[g] + rv[1][g]);
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v0), __float_as_uint(v1), false, false);
                acc[mi][0][g] = __uint_as_float(sw[0]);
                acc[mi][1][g] = __uint_as_float(sw[1]);
            }
            if (full) {
                if (qv) {
                    const unsigned qo = 4u * (unsigned)qb;
#pragma unroll
                    for (int g = 0; g < 16; ++g) {
                        float *orow = yf + (int64_t)(mi * 32 + (g & 3) + 8 * (g >> 2)) * P;
                        st_lane(orow, qo, acc[mi][0][g]);
                        st_lane(orow + 4 * P, qo, acc[mi][1][g]);
                    }
                }
            } else {
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const int row0 = mi * 32 + (g & 3) + 8 * (g >> 2);
                    if (qv && row0 < p.t.C) yf[(int64_t)row0 * P + qb] = acc[mi][0][g];
                    if (qv && row0 + 4 < p.t.C) yf[(int64_t)(row0 + 4) * P + qb] = acc[mi][1][g];
                }
            }
        }
    }
    // ---------------- hand-off: the new y frames become visible to every wave of this workgroup
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");         // this wave's y stores have retired
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");             // drop L1 lines that were pulled before the stores
    // ---------------- phase T
    tcn_step_tile<64, 4, 1, false>(p.t, (int)blockIdx.x, 0, 0, smem);
}

// split-K reduction: out_j[co][p] = ReLU?( sum_ks part[j*ksplit + ks][co][p] (fixed order) + bias[co] + identity residual )
__global__ __launch_bounds__(256) void step_reduce_kernel(const StepParams p) {
    const int64_t P = p.P, P4 = P / 4;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;            // (co, p4)
    if (i >= (int64_t)p.Cout * P4) return;
    const int co = (int)(i / P4);
    const int64_t q = (i - (int64_t)co * P4) * 4;
    const int j = blockIdx.y;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int ks = 0; ks < p.ksplit; ++ks)
        s += *reinterpret_cast<const f32x4 *>(p.part + ((int64_t)(j * p.ksplit + ks) * p.Cout + co) * P + q);
    s += p.bias[co];
    if (p.res_mode == CSK_RES_IDENTITY) {
        const float *xres = p.xres + (int64_t)((p.xres_slot0 + j * p.xres_step) % p.xres_slots) * p.Cres * P;
        s += *reinterpret_cast<const f32x4 *>(xres + (int64_t)co * P + q);
    }
    if (p.relu) { s[0] = relu_nan(s[0]); s[1] = relu_nan(s[1]); s[2] = relu_nan(s[2]); s[3] = relu_nan(s[3]); }
    float *out = p.out + (int64_t)((p.out_slot0 + j) % p.out_slots) * p.Cout * P;
    *reinterpret_cast<f32x4 *>(out + (int64_t)co * P + q) = s;
}

// feat[n, c] = mean over the M*V positions of stream n in a channel-major frame h (C, P): one wave per (n, c)
__global__ __launch_bounds__(256) void co_spatial_pool_kernel(const float *__restrict__ h, float *__restrict__ feat,
                                                              int N, int C, int MV, int64_t P) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;
    if (row >= (int64_t)N * C) return;
    const int n = row / C, c = row % C;
    const float *src = h + (int64_t)c * P + (int64_t)n * MV;
    float s = 0.f;
    for (int j = lane; j < MV; j += 64) s += src[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) feat[row] = s / (float)MV;
}

// pooled[i] = (1/window) * sum over `count` valid ring entries (zero-initialised window, fixed divisor:
// AvgPool1d count_include_pad semantics); entries are visited oldest -> newest so the sum order is fixed
__global__ void co_window_mean_kernel(const float *__restrict__ ring, float *__restrict__ pooled, int64_t n_elem,
                                      int window, int head, int count) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n_elem) return;
    float s = 0.f;
    for (int j = count - 1; j >= 0; --j) {
        int slot = (head - j) % window;
        if (slot < 0) slot += window;
        s += ring[(int64_t)slot * n_elem + i];
    }
    pooled[i] = s / (float)window;
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" int csk_tcn_step_f32(const float *ring, int slots, int head, int head_step, int n_emit, const float *w,
                                const float *x_res, int x_res_slots, int x_res_slot0, int x_res_step,
                                const float *w_res, const float *bias, float *out, int out_slots, int out_slot0,
                                int c, int c_out, int64_t P, int k, int res_mode, int c_res, int relu, int ksplit,
                                float *partial, void *stream) {
    if (!ring || !w || !bias || !out) CSK_FAIL("tcn_step: null pointer");
    if (ksplit < 1 || ksplit > 32 || (ksplit > 1 && !partial)) CSK_FAIL("tcn_step: ksplit must be in [1, 32] and needs a partial-sum buffer");
    if (c <= 0 || c_out <= 0 || P < 4 || (P & 3)) CSK_FAIL("tcn_step: bad dims (P must be a positive multiple of 4)");
    if (k < 1 || k > 9 || slots < k || head < 0 || head >= slots) CSK_FAIL("tcn_step: bad k/slots/head");
    if (n_emit < 1 || n_emit > 64 || head_step < 0 || out_slots < n_emit || out_slot0 < 0 || out_slot0 >= out_slots)
        CSK_FAIL("tcn_step: bad emission geometry");
    if (slots < k - 1 + (n_emit - 1) * head_step + 1) CSK_FAIL("tcn_step: ring too shallow for %d emissions", n_emit);
    if (P >= (1ll << 31) - 256) CSK_FAIL("tcn_step: P too large");
    if (res_mode != CSK_RES_NONE) {
        if (!x_res) CSK_FAIL("tcn_step: residual requested without x_res");
        if (x_res_slots < 1 || x_res_slot0 < 0 || x_res_slot0 >= x_res_slots || x_res_step < 0) CSK_FAIL("tcn_step: bad residual ring geometry");
        if (res_mode == CSK_RES_IDENTITY && c_res != c_out) CSK_FAIL("tcn_step: identity residual needs c_res == c_out");
        if (res_mode == CSK_RES_CONV && !w_res) CSK_FAIL("tcn_step: conv residual without w_res");
    }
    if (((uintptr_t)ring | (uintptr_t)(x_res ? x_res : ring)) & 15) CSK_FAIL("tcn_step: state pointers must be 16-byte aligned");
    StepParams p;
    p.stagger = 0;
    p.stamps = nullptr;
    p.ring = ring; p.w = w; p.xres = x_res ? x_res : ring; p.wres = w_res; p.bias = bias; p.out = out;
    p.C = c; p.Cpad = round_up(c, CSK_CPAD); p.Cout = c_out; p.Mpad = round_up(c_out, CSK_MT);
    p.K = k; p.slots = slots; p.head = head; p.head_step = head_step; p.res_mode = res_mode;
    p.Cres = c_res > 0 ? c_res : 1; p.CresPad = round_up(p.Cres, CSK_CPAD); p.relu = relu; p.P = P;
    // 32-bit lane byte offsets: 4 * (4 * row_stride + position) must stay below 2^32
    p.fast_epi = P < (1ll << 27) && !csk_diag_flag("CSK_SLOW_EPI");
    p.xres_slots = x_res ? x_res_slots : 1; p.xres_slot0 = x_res ? x_res_slot0 : 0; p.xres_step = x_res_step;
    p.out_slots = out_slots; p.out_slot0 = out_slot0;
    // split-K: every split owns >= 1 real channel; fewer splits than asked for if the channel count does not allow more
    p.cper = round_up((p.Cpad + ksplit - 1) / ksplit, KC);
    p.ksplit = ksplit > 1 ? (c + p.cper - 1) / p.cper : 1;
    p.part = partial;
    if (p.ksplit > 1 && ((uintptr_t)partial & 15)) CSK_FAIL("tcn_step: partial-sum buffer must be 16-byte aligned");
    // the slot-balanced 16x16x4 tiles (step16.hip) where the launch shape packs the resident workgroup slots better with
    // them; bitwise the same results (-2: not taken)
    if (p.ksplit == 1) {
        const int rc = csk_launch_tcn_step16(p, n_emit, stream);
        if (rc != -2) return rc;
    }
    const bool big = (p.Mpad % 128) == 0;
    const int MT = big ? 128 : 64, NT = 16384 / MT;
    // emissions folded into a workgroup tile: the largest E in {4, 2, 1} that divides n_emit and leaves >= 64 positions
    // per emission (a wave's 64 columns must belong to one emission); stride-2 launches only as 128-row tiles of two
    // emissions (the only stride-2 shapes of the ST-GCN stack); split-K and the E = 1 form use the plain tile.
    int E = 1;
    if (k == 9 && !csk_diag_flag("CSK_STEP_NOFOLD")) {
        if (!big && head_step == 1 && p.ksplit == 1) E = (n_emit % 4 == 0) ? 4 : (n_emit % 2 == 0) ? 2 : 1;
        if (big && head_step <= 2 && head_step >= 1) E = (n_emit % 2 == 0) ? 2 : 1;      // also with split-K
    }
    const int NP = NT / E;
    void (*kern)(StepParams);
    size_t stage_floats;
    const bool k9 = k == 9 && !csk_diag_flag("CSK_STEP_NOCT");     // E > 1 implies k == 9 (folding condition above)
#define CSK_PICK(MT_, E_, HS_, SP_) (kern = (E_ > 1 || k9) ? tcn_step_kernel<MT_, E_, HS_, SP_, true> : tcn_step_kernel<MT_, E_, HS_, SP_, (E_ > 1)>, stage_floats = RingStage<16384 / MT_ / E_, 8 + (E_ - 1) * HS_ + 1>::LDS_FLOATS)
    if (p.ksplit > 1) {
        if (big) E == 2 ? (head_step == 2 ? CSK_PICK(128, 2, 2, true) : CSK_PICK(128, 2, 1, true)) : CSK_PICK(128, 1, 1, true);
        else CSK_PICK(64, 1, 1, true);
    } else if (big) E == 2 ? (head_step == 2 ? CSK_PICK(128, 2, 2, false) : CSK_PICK(128, 2, 1, false)) : CSK_PICK(128, 1, 1, false);
    else E == 4 ? CSK_PICK(64, 4, 1, false) : E == 2 ? CSK_PICK(64, 2, 1, false) : CSK_PICK(64, 1, 1, false);
#undef CSK_PICK
    const size_t lds = (size_t)(9 * KC * MT + stage_floats) * sizeof(float);
    p.gx = (unsigned)((P + NP - 1) / NP); p.gy = (unsigned)(p.Mpad / MT); p.gz = (unsigned)((n_emit / E) * p.ksplit);
    if ((int64_t)p.gx * p.gy * p.gz >= (1ll << 31)) CSK_FAIL("tcn_step: grid too large");
    dim3 grid(p.gx * p.gy * p.gz);
    // The 64-row, four-emission form also exists within 168 registers (three workgroups per CU, 43 KB of LDS each): taken when
    // the launch needs fewer rounds of 768 resident workgroups than of 512 (256 CUs) -- CoAGCN at the Kinetics shape is 576
    // tiles per 64-channel block: one round instead of a full one plus an eighth.  A function of the launch size only.
    if (!big && E == 4 && p.ksplit == 1 && k9 && !csk_diag_flag("CSK_STEP_NOOCC3")) {
        const unsigned g = grid.x;
        if ((g + 767) / 768 < (g + 511) / 512) kern = tcn_step_kernel<64, 4, 1, false, true, 3>;
    }
    hipStream_t s = (hipStream_t)stream;
    if (const int e = csk_ensure_lds((const void *)kern, lds)) return e;
    hipLaunchKernelGGL(kern, grid, dim3(NTHREADS), lds, s, p);
    if (p.ksplit > 1) {
        if (const int e = (int)hipGetLastError()) return e;
        const int64_t work = (int64_t)c_out * (P / 4);
        hipLaunchKernelGGL(step_reduce_kernel, dim3((unsigned)((work + 255) / 256), n_emit), dim3(256), 0, s, p);
    }
    return (int)hipGetLastError();
}

extern "C" int csk_co_block_step_f32(const float *xin, int xin_slots, int xin_slot0, int c_in, const float *gcn_w,
                                     const float *gcn_bias, const int32_t *ell_src, const float *ell_val,
                                     const int32_t *ell_cnt, int ell_w, int gcn_res_mode, float *y_ring, int y_slots,
                                     int y_slot0, const float *tcn_w, const float *tcn_bias, int res_mode, int x_res_slot0,
                                     float *out, int out_slots, int out_slot0, int c_out, int n_skel, int V, int64_t P,
                                     void *stream) {
    if (!xin || !gcn_w || !gcn_bias || !ell_src || !ell_val || !ell_cnt || !y_ring || !tcn_w || !tcn_bias || !out)
        CSK_FAIL("co_block_step: null pointer");
    if (c_in <= 0 || c_out <= 0 || c_out > 64 || n_skel <= 0 || V < 2 || V > 64 || P < (int64_t)n_skel * V || (P & 3) || P < 4)
        CSK_FAIL("co_block_step: bad dims (c_out <= 64, P a multiple of 4 holding n_skel * V positions)");
    if (P >= (1ll << 31) - 256) CSK_FAIL("co_block_step: P too large");
    if (xin_slots < 8 || y_slots < 12 || out_slots < 4) CSK_FAIL("co_block_step: rings too shallow for a 4-frame cycle");
    if (xin_slot0 < 0 || xin_slot0 >= xin_slots || y_slot0 < 0 || y_slot0 >= y_slots || out_slot0 < 0 || out_slot0 >= out_slots ||
        x_res_slot0 < 0 || x_res_slot0 >= xin_slots)
        CSK_FAIL("co_block_step: slot index out of range");
    if (gcn_res_mode != CSK_RES_IDENTITY && gcn_res_mode != CSK_RES_CONV) CSK_FAIL("co_block_step: gcn_res_mode must be identity or conv");
    if (gcn_res_mode == CSK_RES_IDENTITY && c_in != c_out) CSK_FAIL("co_block_step: identity gcn residual needs c_in == c_out");
    if (res_mode != CSK_RES_NONE && res_mode != CSK_RES_IDENTITY) CSK_FAIL("co_block_step: block residual must be none or identity");
    if (res_mode == CSK_RES_IDENTITY && c_in != c_out) CSK_FAIL("co_block_step: identity block residual needs c_in == c_out");
    if (ell_w < 1 || ell_w > V || ell_cnt[0] < 0 || ell_cnt[0] > 1 || ell_cnt[1] < 0 || ell_cnt[1] > 1 || ell_cnt[2] < 0 || ell_cnt[2] > 4 ||
        ell_cnt[2] > ell_w)
        CSK_FAIL("co_block_step: needs a skeleton-sparse adjacency (<= 1/1/4 non-zeros per column)");
    if (((uintptr_t)xin | (uintptr_t)y_ring | (uintptr_t)out) & 15) CSK_FAIL("co_block_step: state pointers must be 16-byte aligned");
    constexpr int NP = 64;
    CoBlockParams p;
    StepParams &t = p.t;
    t.ring = y_ring; t.w = tcn_w; t.xres = xin; t.wres = nullptr; t.bias = tcn_bias; t.out = out;
    t.C = c_out; t.Cpad = round_up(c_out, CSK_CPAD); t.Cout = c_out; t.Mpad = round_up(c_out, CSK_MT);
    t.K = 9; t.slots = y_slots; t.head = y_slot0; t.head_step = 1; t.res_mode = res_mode;
    t.Cres = res_mode ? c_in : 1; t.CresPad = round_up(t.Cres, CSK_CPAD); t.relu = 1; t.P = P;
    t.fast_epi = P < (1ll << 27) && !csk_diag_flag("CSK_SLOW_EPI");
    t.xres_slots = xin_slots; t.xres_slot0 = x_res_slot0; t.xres_step = 1; t.out_slots = out_slots; t.out_slot0 = out_slot0;
    t.ksplit = 1; t.cper = t.Cpad; t.part = nullptr; t.gx = (unsigned)((P + NP - 1) / NP); t.gy = 1; t.gz = 1;
    p.xin = xin; p.gw = gcn_w; p.gbias = gcn_bias; p.ell_src = ell_src; p.ell_val = ell_val;
    for (int i = 0; i < 3; ++i) p.ell_cnt[i] = ell_cnt[i];
    p.ell_w = ell_w; p.xin_slots = xin_slots; p.xin_slot0 = xin_slot0; p.Cin = c_in; p.CinPad = round_up(c_in, CSK_CPAD);
    p.V = V; p.n_skel = n_skel; p.vmagic = vmagic_of(V); p.fast_epi_g = t.fast_epi;
    p.ldbx = round_up(((NP + V - 2) / V + 1) * V, 4);
    if (p.ldbx > 128) CSK_FAIL("co_block_step: %d joints per skeleton make the input strip of a 64-position tile longer than 128", V);
    // With the slot-balanced tile family (step16.hip) the cycle is that family's fused launch or its two launches: the family's
    // temporal step has its own fp32 summation order, and a block must give the same bits whether its frames arrive one by one
    // or four at a time
    if (csk_step16_enabled()) {
        csk_co_block_args a;
        a.xin = xin; a.xin_slots = xin_slots; a.xin_slot0 = xin_slot0; a.c_in = c_in; a.gcn_w = gcn_w; a.gcn_bias = gcn_bias;
        a.ell_src = ell_src; a.ell_val = ell_val; a.ell_cnt[0] = ell_cnt[0]; a.ell_cnt[1] = ell_cnt[1]; a.ell_cnt[2] = ell_cnt[2];
        a.ell_w = ell_w; a.gcn_res_mode = gcn_res_mode; a.y_ring = y_ring; a.y_slots = y_slots; a.y_slot0 = y_slot0; a.tcn_w = tcn_w;
        a.tcn_bias = tcn_bias; a.res_mode = res_mode; a.x_res_slot0 = x_res_slot0; a.out = out; a.out_slots = out_slots;
        a.out_slot0 = out_slot0; a.c_out = c_out;
        const int rc1 = csk_launch_co_stack16(1, &a, n_skel, V, P, stream);      // one launch where the tile family covers the shape
        if (rc1 != -2) return rc1;
        for (int f = 0; f < 4;) {                              // one graph-conv launch per non-wrapping slot run
            const int xs = (xin_slot0 + f) % xin_slots, ys = (y_slot0 + f) % y_slots;
            int run = 4 - f;
            if (run > xin_slots - xs) run = xin_slots - xs;
            if (run > y_slots - ys) run = y_slots - ys;
            if (const int rc = csk_gcn_stage_f32(xin + (int64_t)xs * c_in * P, y_ring + (int64_t)ys * c_out * P, gcn_w, gcn_bias, ell_src,
                                                 ell_val, ell_cnt, ell_w, 0, 0, run, c_in, c_out, n_skel, V, (int64_t)c_in * P, P,
                                                 (int64_t)c_out * P, P, gcn_res_mode, stream))
                return rc;
            f += run;
        }
        return csk_tcn_step_f32(y_ring, y_slots, y_slot0, 1, 4, tcn_w, res_mode ? xin : nullptr, xin_slots, x_res_slot0, 1, nullptr,
                                tcn_bias, out, out_slots, out_slot0, c_out, c_out, P, 9, res_mode, res_mode ? c_in : 0, 1, 1, nullptr,
                                stream);
    }
    const int R = gcn_res_mode == CSK_RES_CONV ? 4 : 3;
    const size_t lds_g = 2 * (size_t)(R * 8 * 64 + 4 * 8 * p.ldbx), lds_t = (size_t)(9 * KC * 64) + RingStage<64, 12>::LDS_FLOATS;
    const size_t lds = (lds_g > lds_t ? lds_g : lds_t) * sizeof(float);
    void (*kern)(CoBlockParams) = R == 4 ? co_block_kernel<true> : co_block_kernel<false>;
    if (const int e = csk_ensure_lds((const void *)kern, lds)) return e;
    hipLaunchKernelGGL(kern, dim3((unsigned)((P + NP - 1) / NP)), dim3(NTHREADS), lds, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}

extern "C" int csk_co_spatial_pool_f32(const float *h, float *feat, int N, int C, int MV, int64_t P, void *stream) {
    if (!h || !feat) CSK_FAIL("co_spatial_pool: null pointer");
    if (N <= 0 || C <= 0 || MV <= 0 || (int64_t)N * MV > P) CSK_FAIL("co_spatial_pool: bad dims");
    const int64_t rows = (int64_t)N * C;
    hipLaunchKernelGGL(co_spatial_pool_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, h,
                       feat, N, C, MV, P);
    return (int)hipGetLastError();
}

extern "C" int csk_co_window_mean_f32(const float *ring, float *pooled, int64_t n_elem, int window, int head,
                                      int count, void *stream) {
    if (!ring || !pooled) CSK_FAIL("co_window_mean: null pointer");
    if (n_elem <= 0 || window <= 0 || head < 0 || head >= window || count < 0 || count > window)
        CSK_FAIL("co_window_mean: bad dims");
    hipLaunchKernelGGL(co_window_mean_kernel, dim3((unsigned)((n_elem + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, ring, pooled, n_elem, window, head, count);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Multi-stream logit fusion + top-k (scripts/multi_stream_eval.py:33-60): fused = left fold of add / maximum over
// up to 4 prediction arrays (N, classes); rank[n] = number of classes scoring strictly higher than the target
// class (top-k hit <=> rank < k).  One wave per sample.
// ------------------------------------------------------------------------------------------------
struct FuseParams {
    const float *preds[4];
    int n_streams, use_max, N, classes;
    int64_t sample_stride, class_stride;    // element strides of the (N, classes[, steps]) arrays
};

__global__ __launch_bounds__(256) void fuse_rank_kernel(const FuseParams p, const int64_t *__restrict__ targets,
                                                        float *__restrict__ fused, int *__restrict__ rank) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * 4 + wave;
    if (n >= p.N) return;
    const int64_t tgt = targets ? targets[n] : -1;
    float tv = 0.f;
    if (tgt >= 0 && tgt < p.classes) {                     // fused score of the target class (same fold order)
        tv = p.preds[0][n * p.sample_stride + tgt * p.class_stride];
        for (int s = 1; s < p.n_streams; ++s) {
            const float v = p.preds[s][n * p.sample_stride + tgt * p.class_stride];
            tv = p.use_max ? __builtin_elementwise_maximum(tv, v) : tv + v;   // NaN-propagating, as np.maximum
        }
    }
    int higher = 0;
    for (int c = lane; c < p.classes; c += 64) {
        float f = p.preds[0][n * p.sample_stride + c * p.class_stride];
        for (int s = 1; s < p.n_streams; ++s) {
            const float v = p.preds[s][n * p.sample_stride + c * p.class_stride];
            f = p.use_max ? __builtin_elementwise_maximum(f, v) : f + v;
        }
        if (fused) fused[n * p.classes + c] = f;
        higher += (f > tv) ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) higher += __shfl_xor(higher, o);
    if (rank && lane == 0) rank[n] = (tgt >= 0 && tgt < p.classes) ? higher : p.classes;
}

extern "C" int csk_fuse_rank_f32(const float *const *preds, int n_streams, int use_max, int N, int classes,
                                 int64_t sample_stride, int64_t class_stride, const int64_t *targets, float *fused,
                                 int *rank, void *stream) {
    if (!preds || n_streams < 1 || n_streams > 4) CSK_FAIL("fuse_rank: 1..4 prediction arrays expected");
    if (N <= 0 || classes <= 0 || (!fused && !rank)) CSK_FAIL("fuse_rank: bad dims / no output requested");
    if (rank && !targets) CSK_FAIL("fuse_rank: rank requested without targets");
    FuseParams p;
    for (int s = 0; s < 4; ++s) {
        p.preds[s] = s < n_streams ? preds[s] : nullptr;
        if (s < n_streams && !preds[s]) CSK_FAIL("fuse_rank: null prediction array");
    }
    p.n_streams = n_streams; p.use_max = use_max; p.N = N; p.classes = classes;
    p.sample_stride = sample_stride; p.class_stride = class_stride;
    hipLaunchKernelGGL(fuse_rank_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p, targets,
                       fused, rank);
    return (int)hipGetLastError();
}
