// step16.hip -- slot-balanced tiles for the continual (frame-by-frame) path: v_mfma_f32_16x16x4_f32 kernels whose workgroup
// tile is 64 output channels x 16*NB columns, NB = 25 (NTU, V = 25) or 18 (Kinetics, V = 18).
//
// Why a second tile family.  One frame of 1024 NTU streams is P = 51 200 positions = 800 x 64: every 32x32x2 tile shape of
// step.hip / gcn.hip cuts a launch into 800 k tiles for 512 (or 768) resident workgroup slots -- 1.56 (1.04) rounds, i.e. a
// last round that is 56 % (4 %) full on EVERY launch of the cycle (profiles/r05_online_1shard.md: 0.49-0.63 of the fp32 MFMA
// peak where the same tiles reach 0.77-0.83 in the many-round clip launches).  P = 512 x 100, and 100 positions x E emissions
// is 25 column blocks of 16 when E = 4 (C_out = 64), 200 x 2 when E = 2 (C_out = 128, two m-tiles), 400 x 1 when E = 1
// (C_out = 256, four m-tiles): with 16-wide MFMA blocks every launch of the cycle is EXACTLY 512 workgroups of equal work =
// one full round at two workgroups per CU (Kinetics: P = 512 x 72, NB = 18).  v_mfma_f32_16x16x4_f32 has the rate of
// v_mfma_f32_32x32x2_f32 (64 FLOP / clk / SIMD, MI355X_MICROARCH.md) and is the same fmaf chain in ascending k
// (tools/microbench/mfma16_order_probe.hip), so the kernels here are BITWISE interchangeable with the ones they stand in
// for: the K loop visits (chunk, tap, channel) in the same order, the epilogue is the same expression.  Which family a
// launch gets is therefore a pure throughput policy of the launch shape (csk_step16_wins below).
//
// Wave tile: a wave owns 16 output channels (wave w: rows 16 w ..) and ALL NB column blocks: 4 NB accumulator registers.
// The MFMA is issued "transposed" -- A = activations (16 positions x 4 channels), B = weights (4 channels x 16 output
// channels) -- so that a lane's 4 accumulator registers are 4 CONSECUTIVE POSITIONS of one output channel: residual loads and
// output stores are 16-byte accesses without a cross-lane transpose.
//
// LDS window: for an input channel the staged ring slots lie BACK TO BACK in one row (slot w at w * NP), so column
// c = j * NP + p of tap r (emission j reads slot j + r) is at  r * NP + c : affine in the column, every fragment read is one
// VGPR base + an immediate offset.  Stride-2 launches (emission j reads slot 2 j + r) keep the even and the odd slots in two
// such runs.  Row strides are = 16 (mod 32) floats: the four k-slots of a fragment read hit disjoint banks.
#include "tile16.h"
#include "gcn_params.h"
#include "step_params.h"

namespace {

template <int NB, int E, int HS>
struct G16 {
    static constexpr int KCH = 4;                       // channels per chunk of the temporal phase = one MFMA k-step per tap
    static constexpr int KR = 8;                        // channels per chunk of the residual-conv phase
    static constexpr int NT = 16 * NB, NP = NT / E, Q = NP / 4;
    static_assert(NT % E == 0 && NP % 4 == 0, "an emission's positions are whole 16-byte quads");
    static constexpr int NS = 8 + (E - 1) * HS + 1;     // window slots (K = 9)
    static constexpr int NEV = (NS + 1) / 2;            // even slots (stride-2 layout)
    static constexpr int ROW = row16(NS * NP);
    static constexpr int ROWR = row16(NT);
    static constexpr int LDW = 80;                      // weight row stride: 64 output channels, 16 (mod 32)
    static constexpr int WSZ = 9 * KCH * LDW;
    static constexpr int LDS_FLOATS = imax(WSZ + KCH * ROW, KR * LDW + KR * ROWR);
    // LDS offset of window slot w inside a channel row; tap r of emission j is at slot_lds(r) + j * NP
    static constexpr int slot_lds(int w) { return HS == 2 ? ((w & 1) ? (NEV + (w >> 1)) * NP : (w >> 1) * NP) : w * NP; }
};

// TAIL: channel counts that are not whole chunks (C % 4, C_res % 8): the chunk that reaches past C is staged with clamped,
// zeroed rows (uniform branches in the K loop); the fast instantiation has none.
template <int NB, int E, int HS, bool TAIL>
__device__ __forceinline__ void tcn16_tile(const StepParams &p, const int bx, const int by, const int bz, float *smem) {
    typedef G16<NB, E, HS> G;
    constexpr int KCH = G::KCH, NP = G::NP, ROW = G::ROW, LDW = G::LDW;
    float *Wl = smem, *Bl = smem + G::WSZ;
    int tid_ = threadIdx.x;
    asm volatile("" : "+v"(tid_));      // opaque: inside a fused stack the per-thread setup must not be hoisted out of the block loop (and spilled)
    const int tid = tid_, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kq = lane >> 4;
    const int m0 = by * 64, p0 = bx * NP, j0 = bz * E;
    const int64_t P = p.P;
    int first = (p.head + j0 * p.head_step - 8) % p.slots;                     // ring slot of window slot 0
    if (first < 0) first += p.slots;

    f32x4 acc[NB];
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) acc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    // diagnostic (p.stamps): per wave the cycles of the three parts of a chunk summed over the K loop, per workgroup start /
    // loop start / loop end / end
    unsigned long long st0 = 0, st1 = 0, st2 = 0, ph0 = 0, ph1 = 0, ph2 = 0, tq = 0;
    if (p.stamps) st0 = __builtin_amdgcn_s_memtime();

    // fragment bases: weights (B operand) lane (n = l15 -> output channel, k = kq), activations (A operand) lane (i = l15 ->
    // position inside the column block, k = kq)
    const float *wl_lane = Wl + kq * LDW + wave * 16 + l15;
    const float *bl_lane = Bl + kq * ROW + l15;
    {   // ---- phase 1: temporal conv over the ring window
        W16<9, KCH, LDW> ws;
        Win16<KCH, G::NS, NP, ROW, HS == 2> rs;
        ws.setup(p.Cpad, p.Mpad, tid);
        rs.setup(first, 1, p.slots, (int64_t)p.C * P, P, (int)(P - 4 - p0), tid);
        const float *wbase = p.w + m0;
        const float *rbase = p.ring + p0;
        ws.issue(wbase);
        if (TAIL && KCH > p.C) rs.issue_tail(rbase, 0, p.C, P, tid);
        else rs.issue(rbase);
        if (p.stamps) st1 = tq = __builtin_amdgcn_s_memtime();
        // Issue priority inside the MFMA segments.  The two workgroups of a CU do identical work; at EQUAL priority the
        // arbiter prefers the older wave: the workgroup that started ~400 cycles earlier runs 22 % ahead, finishes, and its
        // partner walks the rest of its K loop alone -- bound by its own LDS round trips (stamps, 256-channel launch: 951 k
        // against 1 215 k cycles in EVERY CU).  So the favoured workgroup alternates chunk by chunk (priority 2 against 1):
        // 1 058 k / 1 153 k, the launch 4.6 % shorter.  p.stagger >> 16 (CSK_TCN16_PRIO under CSK_DIAG=1): 0 = equal
        // priorities, 1 = alternating, 2 = always the younger workgroup.
        const int pmode = p.stagger >> 16;
        int turn = pmode == 0 ? 0 : (int)(__builtin_amdgcn_s_getreg(6148) & 1);   // HW_ID wave slot: 0 = the older workgroup of the CU
        for (int c0 = 0; c0 < p.Cpad; c0 += KCH) {
            __syncthreads();
            if (p.stamps) { const unsigned long long t = __builtin_amdgcn_s_memtime(); ph0 += t - tq; tq = t; }
            ws.commit(Wl);
            rs.commit(Bl);
            __syncthreads();
            if (p.stamps) { const unsigned long long t = __builtin_amdgcn_s_memtime(); ph1 += t - tq; tq = t; }
            // next chunk's loads in three bursts between the three tap segments; past the end the last chunk is loaded again
            // into the (then dead) staging registers so that the K loop stays one basic block
            const int cn = min(c0 + KCH, p.Cpad - KCH);
            const float *wnext = wbase + (size_t)cn * p.Mpad, *rnext = rbase + (int64_t)cn * P;
            const bool tail = TAIL && cn + KCH > p.C;                           // uniform
            ws.issue_one(0, wnext);
            if (!tail) rs.template issue_third<0>(rnext);
            if (turn & 1) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1);
            turn += pmode & 1;
#pragma unroll
            for (int r = 0; r < 3; ++r) mfma16_tap<NB>(wl_lane + r * KCH * LDW, bl_lane + G::slot_lds(r), acc);
            ws.issue_one(1, wnext);
            if (!tail) rs.template issue_third<1>(rnext);
#pragma unroll
            for (int r = 3; r < 6; ++r) mfma16_tap<NB>(wl_lane + r * KCH * LDW, bl_lane + G::slot_lds(r), acc);
            ws.issue_one(2, wnext);
            if (!tail) rs.template issue_third<2>(rnext);
            else rs.issue_tail(rnext, cn, p.C, P, tid);
#pragma unroll
            for (int r = 6; r < 9; ++r) mfma16_tap<NB>(wl_lane + r * KCH * LDW, bl_lane + G::slot_lds(r), acc);
            __builtin_amdgcn_s_setprio(0);
            if (p.stamps) { const unsigned long long t = __builtin_amdgcn_s_memtime(); ph2 += t - tq; tq = t; }
        }
        if (p.stamps) st2 = __builtin_amdgcn_s_memtime();
    }
    // ---- phase 2: 1x1 residual conv on the delayed block input (one "tap" per emission: the E residual frames back to back)
    if (p.res_mode == CSK_RES_CONV) {
        constexpr int KR = G::KR, ROWR = G::ROWR;
        float *Wr = smem, *Br = smem + KR * LDW;
        W16<1, KR, LDW> ws;
        Win16<KR, E, NP, ROWR, false> rs;
        ws.setup(p.CresPad, p.Mpad, tid);
        rs.setup((p.xres_slot0 + j0 * p.xres_step) % p.xres_slots, p.xres_step, p.xres_slots, (int64_t)p.Cres * P, P, (int)(P - 4 - p0), tid);
        const float *wbase = p.wres + m0;
        const float *xbase = p.xres + p0;
        const float *wr_lane = Wr + kq * LDW + wave * 16 + l15;
        const float *br_lane = Br + kq * ROWR + l15;
        ws.issue(wbase);
        if (TAIL && KR > p.Cres) rs.issue_tail(xbase, 0, p.Cres, P, tid);
        else rs.issue(xbase);
        for (int c0 = 0; c0 < p.CresPad; c0 += KR) {
            __syncthreads();
            ws.commit(Wr);
            rs.commit(Br);
            __syncthreads();
            const int cn = min(c0 + KR, p.CresPad - KR);
            ws.issue(wbase + (size_t)cn * p.Mpad);
            if (TAIL && cn + KR > p.Cres) rs.issue_tail(xbase + (int64_t)cn * P, cn, p.Cres, P, tid);
            else rs.issue(xbase + (int64_t)cn * P);
#pragma unroll
            for (int s = 0; s < KR / 4; ++s) mfma16_tap<NB>(wr_lane + 4 * s * LDW, br_lane + 4 * s * ROWR, acc);
        }
    }
    // ---- epilogue: + bias (+ identity residual), ReLU
    unsigned oslot[E], xslot[E];
#pragma unroll
    for (int j = 0; j < E; ++j) {
        oslot[j] = (unsigned)((int64_t)((p.out_slot0 + j0 + j) % p.out_slots) * p.Cout * P * 4);
        xslot[j] = (unsigned)((int64_t)((p.xres_slot0 + (j0 + j) * p.xres_step) % p.xres_slots) * p.Cres * P * 4);
    }
    const int nval = (int)min((int64_t)NP, P - p0);                             // positions of the tile inside the row (multiple of 4)
    epilogue16<NB, E, NP>(acc, p.bias, p.Cout, m0 + wave * 16 + l15, kq, p.res_mode == CSK_RES_IDENTITY, p.relu != 0, p.xres, p.out,
                          xslot, oslot, P, P, p0, nval, nval);
    if (p.stamps && lane == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // the epilogue's stores have left
        unsigned long long *o = p.stamps + ((size_t)blockIdx.x * 4 + wave) * 8;
        o[0] = st0; o[1] = st1; o[2] = st2; o[3] = __builtin_amdgcn_s_memtime(); o[4] = ph0; o[5] = ph1; o[6] = ph2;
        o[7] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));    // HW_REG_HW_ID
    }
}

template <int NB, int E, int HS, bool TAIL>
__global__ __launch_bounds__(NTHREADS, TAIL ? 1 : 2) void tcn_step16_kernel(const StepParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    stagger_odd_slot(p.stagger & 0xffff);
    // XCD-contiguous work order, m-tile fastest (the m-tiles of a position tile read the same ring window)
    const unsigned wid = xcd_contiguous_id(blockIdx.x, gridDim.x);
    tcn16_tile<NB, E, HS, TAIL>(p, (int)(wid / (p.gy * p.gz)), (int)(wid % p.gy), (int)((wid / p.gy) % p.gz), smem);
}

// ------------------------------------------------------------------------------------------------
// Graph conv on the same tiles (csk_gcn_stage_f32 for skeleton-sparse adjacencies, <= 1 / 1 / 4 non-zeros per column):
//     y[f][co][q] = ReLU( sum_r sum_c W_r[c][co] * agg_r(x_f)[c][q] + bias[co] + gcn_residual )
// Tile = 64 output channels x (F segments x NPG positions), NPG a multiple of V: a tile holds WHOLE skeletons, so the raw input
// chunk of the tile's own columns is all the aggregation needs.  Per 8-channel chunk:
//   P1  every thread aggregates its <= 2 columns for the 8 channels from the staged x rows (adjacency entries in registers,
//       the expression of gcn_stage_sparse2_kernel) into LDS rows in the K ORDER of that kernel -- (channel pair s, subset r,
//       channel 2 s + h) -> row (s R + r) 2 + h -- and commits the chunk's weights in the same row order;
//   P2  the MFMAs walk the rows four at a time (the chain of the 32x32x2 kernel's k-step pairs: bitwise the same sums); the
//       next chunk's x rows are committed and the one after is loaded underneath.
// Two barriers per chunk, x single-buffered (read in P1, rewritten in P2).
// Segment f of a group is slot (ring_slot0 + seg0 + f) % ring_slots of x / y (a plain csk_gcn_stage_f32 call: no wrap).  A channel
// count that is not a multiple of 8: the chunk holding the last real channels re-reads the last real row for the rows past it,
// chunks of padding only re-read that chunk (their packed weights are zero; the value only has to be finite -- the clamp of
// gcn_stage_sparse2_kernel).
#ifndef CSK_GCN16_WAVES
#define CSK_GCN16_WAVES 8      // waves per workgroup of the stand-alone graph-conv launches (4: the round's first form, A/B builds)
#endif
#ifndef CSK_GCN_AHEAD
#define CSK_GCN_AHEAD CSK_READ_AHEAD
#endif
constexpr int GCN_AHEAD = CSK_GCN_AHEAD;
// NW = 4: a wave owns 16 output channels x all NB column blocks (the form the fused stack runs: 256 threads, shared with the
// temporal step).  NW = 8 (the stand-alone launches): 512 threads, a wave owns 16 channels x HALF the column blocks -- 52
// accumulator registers, <= 128 VGPRs, FOUR waves per SIMD at two workgroups per CU: both phases of this kernel are latency chains
// (LDS gathers -> arithmetic -> LDS writes; operand read -> MFMA), and twice the waves hide twice the latency.
template <int NB, int F, bool CONVRES, int NW = 4>
__device__ __forceinline__ void gcn16_tile(const GcnParams &p, const int mt, const int qt, const int sg, float *smem) {
    constexpr int R = CONVRES ? 4 : 3, KCG = 8, NE = KCG * R;
    constexpr int NT = 16 * NB, NPG = NT / F, AROW = row16(NT), LDW = 80;
    constexpr int NTH = 64 * NW, NBW = NW == 8 ? (NB + 1) / 2 : NB;    // threads; column blocks of a wave
    constexpr int NC = NW == 8 ? 1 : 2;                                // columns a thread aggregates (adjacent ones)
    typedef WinT16<F, NPG, true, NTH> XS;                              // x rows staged channel-interleaved (tile16.h)
    constexpr int XH = XS::HALF;
    static_assert(NW == 4 || NW == 8, "four or eight waves");
    static_assert(NT % NC == 0 && NT <= NC * NTH, "every column has a thread");
    static_assert(NT % F == 0 && NPG % 4 == 0, "a segment's positions are whole 16-byte quads");
    float *Wl = smem, *Ba = smem + NE * LDW, *Xs = Ba + NE * AROW;
    int tid_ = threadIdx.x;
    asm volatile("" : "+v"(tid_));      // opaque: inside a fused stack the per-thread setup must not be hoisted out of the block loop (and spilled)
    const int tid = tid_, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kq = lane >> 4;
    const int wm = wave & 3, wh = wave >> 2;                           // 16-channel block; half of the column blocks (NW = 8)
    const int m0 = mt * 64, q0 = qt * NPG, seg0 = sg * F;
    const int V = p.V, Q = p.frames * V;
    const int nval = min(NPG, Q - q0);                                 // valid positions of the tile: whole skeletons
    f32x4 acc[NBW];
#pragma unroll
    for (int cb = 0; cb < NBW; ++cb) acc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};

    W16<R, KCG, LDW, true, NTH> ws;
    XS xs;
    const int nfull = p.Cin / KCG, rem = p.Cin % KCG;                  // whole chunks, real channels of the partial one
    const int clast = rem ? nfull : nfull - 1;                         // last chunk with a real channel
    ws.setup(p.CinPad, p.Mpad, tid);
    xs.setup(p.x_ring_slot0 + seg0, 1, p.x_ring_slots, p.x_seg_stride, p.x_chan_stride, (int)(p.x_chan_stride - 4 - q0), tid, rem ? rem : KCG);
    const float *wbase = p.w + m0;
    const float *xbase = p.x + q0;
    const float *wl_lane = Wl + kq * LDW + wm * 16 + l15;
    // (NW = 8, NB odd: the second half's last block lies past the tile: not computed, masked by the epilogue)
    const float *ba_lane = Ba + kq * AROW + l15 + 16 * NBW * wh;
    const int nchunks = p.CinPad / KCG;
    auto issue_x = [&](int c) {                                        // chunk c of the K loop
        const int cc = min(c, clast);
        xs.issue_sel(xbase + (int64_t)cc * KCG * p.x_chan_stride, rem != 0 && cc == nfull);
    };
    ws.issue(wbase);
    issue_x(0);
    // (behind the first chunk's loads, whose latency covers the table reads)
    // adjacency entries of this thread's two columns c0, c0 + 1 (offsets inside the interleaved x tile); threads past the tile
    // redo its last pair (the same values to the same addresses), a wave with no column of its own skips the phase
    int eoff[NC][6], ioff[NC];
    float eval[NC][6];
    const int c0 = min(NC * tid, NT - NC);
    const bool p1_wave = wave * 64 * NC < NT;                          // (wave-uniform)
#pragma unroll
    for (int n = 0; n < NC; ++n) {
        const int col = c0 + n;
        const int f = col / NPG, pos = min(col - f * NPG, nval - 1);
        const int t = div_magic(pos, p.vmagic), w = pos - t * V, sb = f * NPG + t * V;
        ioff[n] = xt_off(sb + w);
#pragma unroll
        for (int e = 0; e < 6; ++e) {
            const int r = e < 2 ? e : 2, k = e < 2 ? 0 : e - 2;          // subsets 0,1: one entry; subset 2: four
            const bool have = k < p.ell_cnt[r];
            const int idx = (r * V + w) * p.ell_w + min(k, p.ell_w - 1);
            eoff[n][e] = xt_off(sb + (have ? p.ell_src[idx] : 0));
            eval[n][e] = have ? p.ell_val[idx] : 0.f;
        }
    }
    __syncthreads();                                                   // (a previous phase of a fused launch may still read LDS)
    xs.commit(Xs);
    issue_x(1);
    __syncthreads();
    const int gmode = (p.stagger >> 20) & 7, godd = (int)(__builtin_amdgcn_s_getreg(6148) & 1);   // HW_ID wave slot: 0 = the older workgroup
    unsigned long long gp0 = 0, gp1 = 0, gp2 = 0, gp3 = 0, gp4 = 0, gq = 0, gst0 = 0;   // diagnostic phase sums (p.stamps only)
    if (p.stamps) gst0 = gq = __builtin_amdgcn_s_memtime();
    for (int c = 0; c < nchunks; ++c) {
        // ---- P1: aggregate chunk c, commit its weights, load the next chunk's.  Four rounds (channel half h, column n): the six
        // source joints of a column with 4 channels each in six 16-byte gathers from the channel-interleaved x tile (WinT16),
        // multiplied out, the two columns of a thread written as pairs
        if (!(p.stagger & 0x10000) && p1_wave) {               // (diagnostic: CSK_GCN16_SKIP=1 times the kernel without its aggregation phase)
#pragma unroll
        for (int h = 0; h < 2; ++h) {                          // channels 4 h .. 4 h + 3 of the chunk
            float res[4][3][NC];
#pragma unroll
            for (int n = 0; n < NC; ++n) {
                // (the six 16-byte gathers of the NEXT round in flight under this round's arithmetic: 4.75 k -> 4.45 k cycles for
                // the phase, which the partner's MFMA phase lost again -- 6.7 k -> 7.3 k: not kept)
                f32x4 x[6];
#pragma unroll
                for (int e = 0; e < 6; ++e) x[e] = *reinterpret_cast<const f32x4 *>(Xs + h * XH + eoff[n][e]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    res[kk][0][n] = eval[n][0] * x[0][kk];
                    res[kk][1][n] = eval[n][1] * x[1][kk];
                    float s2 = eval[n][2] * x[2][kk];
                    s2 = fmaf(eval[n][3], x[3][kk], s2);
                    s2 = fmaf(eval[n][4], x[4][kk], s2);
                    s2 = fmaf(eval[n][5], x[5][kk], s2);
                    res[kk][2][n] = s2;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    float *d = Ba + gcn_entry(4 * h + kk, r, R) * AROW + c0;
                    if constexpr (NC == 2) *reinterpret_cast<f32x2 *>(d) = f32x2{res[kk][r][0], res[kk][r][NC - 1]};
                    else *d = res[kk][r][0];
                }
            if (CONVRES) {                                    // fourth "subset": the input itself (the conv gcn_residual's operand)
                const f32x4 x0 = *reinterpret_cast<const f32x4 *>(Xs + h * XH + ioff[0]);
                const f32x4 x1 = *reinterpret_cast<const f32x4 *>(Xs + h * XH + ioff[NC - 1]);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    float *d = Ba + gcn_entry(4 * h + kk, 3, R) * AROW + c0;
                    if constexpr (NC == 2) *reinterpret_cast<f32x2 *>(d) = f32x2{x0[kk], x1[kk]};
                    else *d = x0[kk];
                }
            }
        }
        }
        ws.commit(Wl);
        ws.issue(wbase + (size_t)min(c + 1, nchunks - 1) * KCG * p.Mpad);
        if (p.stamps) { const unsigned long long t = __builtin_amdgcn_s_memtime(); gp0 += t - gq; gq = t; }
        __syncthreads();
        if (p.stamps) { const unsigned long long t = __builtin_amdgcn_s_memtime(); gp1 += t - gq; gq = t; }
        // ---- P2: x rows of chunk c + 1 -> LDS, chunk c + 2 -> registers, MFMAs of chunk c
        xs.commit(Xs);
        issue_x(c + 2);
        if (p.stamps) { const unsigned long long t = __builtin_amdgcn_s_memtime(); gp2 += t - gq; gq = t; }
        // (issue priority of the MFMA phase: see tcn16_tile -- mode 1 / 2: the favoured workgroup of the CU alternates chunk by chunk)
        if (gmode == 0) __builtin_amdgcn_s_setprio(1);
        else if (gmode != 4) { if (((gmode == 3 ? 0 : c) + godd) & 1) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1); }
        if (!(p.stagger & 0x20000)) {                          // (diagnostic: CSK_GCN16_SKIP=2: without its MFMA phase)
#pragma unroll
        for (int m = 0; m < NE / 4; ++m) {
            if constexpr (NW == 8 && (NB & 1)) mfma16_tap_opt_last<NBW, GCN_AHEAD>(wl_lane + 4 * m * LDW, ba_lane + 4 * m * AROW, acc, wh == 0);
            else mfma16_tap<NBW, GCN_AHEAD>(wl_lane + 4 * m * LDW, ba_lane + 4 * m * AROW, acc);
        }
        }
        __builtin_amdgcn_s_setprio(0);
        if (p.stamps) { const unsigned long long t = __builtin_amdgcn_s_memtime(); gp3 += t - gq; gq = t; }
        __syncthreads();
        if (p.stamps) { const unsigned long long t = __builtin_amdgcn_s_memtime(); gp4 += t - gq; gq = t; }
    }
    if (p.stamps && lane == 0 && wave < 4) {
        unsigned long long *o = p.stamps + ((size_t)blockIdx.x * 4 + wave) * 8;
        o[0] = gst0; o[1] = gq; o[2] = gp0; o[3] = gp1; o[4] = gp2; o[5] = gp3; o[6] = gp4;
        o[7] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));    // HW_REG_HW_ID
    }
    unsigned oslot[F], xslot[F];
#pragma unroll
    for (int f = 0; f < F; ++f) {
        oslot[f] = (unsigned)((int64_t)((p.y_ring_slot0 + seg0 + f) % p.y_ring_slots) * p.y_seg_stride * 4);
        xslot[f] = (unsigned)((int64_t)((p.x_ring_slot0 + seg0 + f) % p.x_ring_slots) * p.x_seg_stride * 4);
    }
    const int nrow = (int)min((int64_t)NPG, min(p.x_chan_stride, p.y_chan_stride) - q0) & ~3;
    // (lane coordinates recomputed from an opaque thread id: kept live across the K loop they would be the registers that spill
    // in the 256-register conv-residual instantiation)
    int tid2 = threadIdx.x;
    asm volatile("" : "+v"(tid2));
    const int wave2 = __builtin_amdgcn_readfirstlane(tid2 >> 6);
    epilogue16<NBW, F, NPG, false, NW == 8 ? 4 : 5>(acc, p.bias, p.Cout, m0 + (wave2 & 3) * 16 + (tid2 & 15), (tid2 & 63) >> 4, !CONVRES, true,
                                                     p.x, p.y, xslot, oslot, p.x_chan_stride, p.y_chan_stride, q0, nrow, nval, 16 * NBW * (wave2 >> 2));
}

template <int NB, int F, bool CONVRES, int NW>
__global__ __launch_bounds__(64 * NW, NW / 2) void gcn16_kernel(const GcnParams p) {   // (second figure: waves per SIMD = two workgroups per CU)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    stagger_odd_slot(p.stagger & 0xffff);
    const unsigned wid = xcd_contiguous_id(blockIdx.x, gridDim.x);
    unsigned long long k0 = 0;
    if (p.stamps) k0 = __builtin_amdgcn_s_memtime();
    gcn16_tile<NB, F, CONVRES, NW>(p, (int)(wid % p.mtiles), (int)((wid / p.mtiles) % p.qtiles), (int)(wid / (p.mtiles * p.qtiles)), smem);
    if (p.stamps && (threadIdx.x & 63) == 0) {                 // diagnostic: workgroup start / end (stores retired), behind the per-wave records
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long *o = p.stamps + (size_t)gridDim.x * 32 + ((size_t)blockIdx.x * 4 + ((threadIdx.x >> 6) & 3)) * 2;
        if ((threadIdx.x >> 6) < 4) { o[0] = k0; o[1] = __builtin_amdgcn_s_memtime(); }
    }
}

// ------------------------------------------------------------------------------------------------
// A stack of 64-channel continual blocks in ONE launch (csk_co_stack_step_f32): the workgroup that owns NP positions of the
// four new frames carries them through graph conv and temporal step of block 1, 2, ... -- every dependency of a tile is on
// the SAME tile of the stage before (step mode has no temporal halo and a tile holds whole skeletons), so no workgroup ever
// waits for another.  A stage's output reaches the next stage through the state rings (it is state: later cycles need it)
// and L2: stores retired, workgroup barrier, L1 invalidated.  Same tile functions as the stand-alone launches: bitwise the
// same results; what the stack saves is launches with their fill / drain and the co-resident workgroups of a CU drifting
// apart into different phases (matrix beside memory instead of matrix beside matrix).
// ------------------------------------------------------------------------------------------------
struct CoStackBlock {
    GcnParams g;
    StepParams t;
};
struct CoStackParams {
    CoStackBlock b[CSK_CO_STACK_MAX];
    int nblk;
};

__device__ __forceinline__ void stage_handoff() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");         // this wave's stores have retired
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");             // drop L1 lines that were pulled before the stores
}

template <int NB>
__global__ __launch_bounds__(NTHREADS, 2) void co_stack16_kernel(const CoStackParams sp) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    stagger_odd_slot(sp.b[0].g.stagger & 0xffff);
    const int bx = (int)xcd_contiguous_id(blockIdx.x, gridDim.x);
    for (int i = 0; i < sp.nblk; ++i) {
        const CoStackBlock &b = sp.b[i];
        gcn16_tile<NB, 4, false>(b.g, 0, bx, 0, smem);         // identity gcn_residual only (c_in == c_out): see csk_launch_co_stack16
        stage_handoff();
        tcn16_tile<NB, 4, 1, false>(b.t, bx, 0, 0, smem);
        if (i + 1 < sp.nblk) stage_handoff();
    }
}


template <int NB, int F>
int launch_gcn16(GcnParams p, int n_seg, hipStream_t s) {
    constexpr int NT = 16 * NB, NPG = NT / F;
    const int Q = p.frames * p.V;
    p.qtiles = (unsigned)((Q + NPG - 1) / NPG); p.mtiles = (unsigned)(p.Mpad / 64);
    const int64_t grid = (int64_t)p.qtiles * p.mtiles * (n_seg / F);
    if (grid >= (1ll << 31)) CSK_FAIL("gcn_stage: grid too large");
    constexpr int NW = CSK_GCN16_WAVES;
    void (*kern)(GcnParams) = p.R == 4 ? gcn16_kernel<NB, F, true, NW> : gcn16_kernel<NB, F, false, NW>;
    p.stagger = stagger_units("CSK_GCN16_STAGGER", GCN16_STAGGER) | ((csk_diag_int("CSK_GCN16_SKIP") & 15) << 16) | (prio_mode("CSK_GCN16_PRIO", GCN16_PRIO) << 20);
    p.stamps = csk_diag_stamps();
    const size_t lds = (size_t)(8 * p.R * (80 + row16(NT)) + 2 * WinT16<F, NPG>::HALF) * sizeof(float);
    if (const int e = csk_ensure_lds((const void *)kern, lds)) return e;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * NW), lds, s, p);
    return (int)hipGetLastError();
}

template <int NB, int E, int HS>
int launch16(StepParams p, int n_emit, hipStream_t s) {
    typedef G16<NB, E, HS> G;
    p.gx = (unsigned)((p.P + G::NP - 1) / G::NP); p.gy = (unsigned)(p.Mpad / 64); p.gz = (unsigned)(n_emit / E);
    if ((int64_t)p.gx * p.gy * p.gz >= (1ll << 31)) CSK_FAIL("tcn_step: grid too large");
    // the fast instantiation walks Cpad (CresPad) channel rows: it is only for operands without padding rows -- a channel count
    // that is a multiple of the chunk but not of CSK_CPAD (C = 4, 8: test shapes) would read up to 12 rows past the last ring slot
    const bool tail = p.Cpad != p.C || (p.res_mode == CSK_RES_CONV && p.CresPad != p.Cres);
    p.stagger = stagger_units("CSK_TCN16_STAGGER", TCN16_STAGGER) | (prio_mode("CSK_TCN16_PRIO", TCN16_PRIO) << 16);
    p.stamps = csk_diag_stamps();
    void (*kern)(StepParams) = tail ? tcn_step16_kernel<NB, E, HS, true> : tcn_step16_kernel<NB, E, HS, false>;
    const size_t lds = (size_t)G::LDS_FLOATS * sizeof(float);
    if (const int e = csk_ensure_lds((const void *)kern, lds)) return e;
    hipLaunchKernelGGL(kern, dim3(p.gx * p.gy * p.gz), dim3(NTHREADS), lds, s, p);
    return (int)hipGetLastError();
}

}  // namespace

// ---- dispatch ------------------------------------------------------------------------------------------------------------
// CSK_STEP16 (under CSK_DIAG=1): 1 = never (the 32x32x2 kernels everywhere: A/B runs).
// Temporal step: the K loop of this family walks (4-channel chunk, tap) where the 32x32x2 kernels walk (8-channel chunk, tap) --
// a different fp32 summation order -- so WHICH family runs must not depend on the launch size (a stream's results must not
// depend on how many streams share the slab, nor on how many frames a launch carries): every k = 9, unsplit launch whose
// rings fit 32-bit byte offsets takes this family; only the tile WIDTH (NB) follows the launch shape.
int csk_step16_enabled() { return csk_diag_int("CSK_STEP16") != 1; }

int csk_launch_tcn_step16(StepParams p, int n_emit, void *stream) {
    if (!csk_step16_enabled()) return -2;
    if (p.K != 9 || p.ksplit != 1 || p.head_step < 1 || p.head_step > 2) return -2;
    // 32-bit byte offsets inside the rings
    const int64_t ring_bytes = (int64_t)p.slots * p.C * p.P * 4, xres_bytes = (int64_t)p.xres_slots * p.Cres * p.P * 4;
    const int64_t out_bytes = (int64_t)p.out_slots * p.Cout * p.P * 4;
    if (ring_bytes >= (1ll << 32) || xres_bytes >= (1ll << 32) || out_bytes >= (1ll << 32) || p.P < 8) return -2;
    const int E = p.head_step == 2 ? (n_emit % 2 == 0 ? 2 : 1) : (n_emit % 4 == 0 ? 4 : n_emit % 2 == 0 ? 2 : 1);
    const int HS = E > 1 ? p.head_step : 1;
    const int64_t mt = p.Mpad / 64;
    int best_nb = 0;
    double best = 0;
    for (int nb : {25, 18}) {
        const int np = 16 * nb / E;
        const double c = cost_model(((p.P + np - 1) / np) * mt * (n_emit / E), 16.0 * nb);
        if (!best_nb || c < best) { best = c; best_nb = nb; }
    }
    hipStream_t s = (hipStream_t)stream;
#define CSK_L16(NB_) (E == 4 ? launch16<NB_, 4, 1>(p, n_emit, s) : E == 2 ? (HS == 2 ? launch16<NB_, 2, 2>(p, n_emit, s) : launch16<NB_, 2, 1>(p, n_emit, s)) : launch16<NB_, 1, 1>(p, n_emit, s))
    return best_nb == 25 ? CSK_L16(25) : CSK_L16(18);
#undef CSK_L16
}

// Graph conv: bitwise the sums of gcn_stage_sparse2_kernel, so the choice IS a throughput policy of the launch shape: taken
// where the cost model says the launch packs the chip better (CSK_GCN16=2 under CSK_DIAG=1: whenever the shape is supported).
int csk_launch_gcn16(GcnParams p, int n_seg, void *stream) {
    const int mode = csk_diag_int("CSK_GCN16");
    if (!csk_step16_enabled() || mode == 1) return -2;
    if (p.adj_seg_stride != 0 || p.ksplit != 1 || p.ell_cnt[0] > 1 || p.ell_cnt[1] > 1 || p.ell_cnt[2] > 4) return -2;
    if ((p.x_seg_stride | p.x_chan_stride | p.y_seg_stride | p.y_chan_stride) & 3) return -2;
    if (((uintptr_t)p.x | (uintptr_t)p.y) & 15) return -2;
    const int F = n_seg % 4 == 0 ? 4 : n_seg % 2 == 0 ? 2 : 1;
    if ((int64_t)F * p.x_seg_stride * 4 >= (1ll << 32) || (int64_t)F * p.y_seg_stride * 4 >= (1ll << 32)) return -2;
    const int Q = p.frames * p.V;
    const int64_t mt = p.Mpad / 64;
    int best_nb = 0;
    double best = 0;
    for (int nb : {25, 18}) {
        const int npg = 16 * nb / F;
        if (npg % p.V) continue;                              // tiles hold whole skeletons
        const double c = cost_model((int64_t)((Q + npg - 1) / npg) * mt * (n_seg / F), 16.0 * nb);
        if (!best_nb || c < best) { best = c; best_nb = nb; }
    }
    if (!best_nb) return -2;
    if (mode != 2) {
        const bool big = (p.Mpad % 128) == 0;
        const int nt32 = big ? 128 : 256;
        const double c32 = cost_model((int64_t)((Q + nt32 - 1) / nt32) * (big ? p.Mpad / 128 : p.Mpad / 64) * n_seg, 256.0);
        if (best >= 0.97 * c32) return -2;                   // not clearly better: keep the 32x32x2 tiles
    }
    hipStream_t s = (hipStream_t)stream;
#define CSK_G16(NB_) (F == 4 ? launch_gcn16<NB_, 4>(p, n_seg, s) : F == 2 ? launch_gcn16<NB_, 2>(p, n_seg, s) : launch_gcn16<NB_, 1>(p, n_seg, s))
    return best_nb == 25 ? CSK_G16(25) : CSK_G16(18);
#undef CSK_G16
}

// ---- stack of 64-channel blocks ------------------------------------------------------------------------------------------
// -2: not a shape the fused stack is built for (the caller issues the per-stage launches); CSK_STACK16=1 under CSK_DIAG=1: never
int csk_launch_co_stack16(int n_blocks, const csk_co_block_args *b, int n_skel, int V, int64_t P, void *stream) {
    if (!csk_step16_enabled() || csk_diag_int("CSK_STACK16") == 1) return -2;
    if (n_blocks < 1 || n_blocks > CSK_CO_STACK_MAX || P < 8 || (P & 3)) return -2;
    const int64_t Q = (int64_t)n_skel * V;
    int best_nb = 0;
    double best = 0;
    for (int nb : {25, 18}) {
        const int np = 16 * nb / 4;
        if (np % V) continue;                                 // tiles hold whole skeletons
        const double c = cost_model((P + np - 1) / np, 16.0 * nb);
        if (!best_nb || c < best) { best = c; best_nb = nb; }
    }
    if (!best_nb) return -2;
    const int NP = 16 * best_nb / 4;
    CoStackParams sp;
    sp.nblk = n_blocks;
    for (int i = 0; i < n_blocks; ++i) {
        const csk_co_block_args &a = b[i];
        // (c_out a multiple of CSK_CPAD: the stack's temporal step is the instantiation without padding rows)
        if (a.c_out > 64 || (a.c_out % CSK_CPAD) || a.ell_cnt[0] > 1 || a.ell_cnt[1] > 1 || a.ell_cnt[2] > 4) return -2;
        // a conv gcn_residual (c_in != c_out: the first block of a network) has a fourth operand subset and with it the register
        // budget of a kernel of its own: such a block runs as its two launches
        if (a.gcn_res_mode != CSK_RES_IDENTITY) return -2;
        if ((int64_t)a.xin_slots * a.c_in * P * 4 >= (1ll << 32) || (int64_t)a.y_slots * a.c_out * P * 4 >= (1ll << 32) ||
            (int64_t)a.out_slots * a.c_out * P * 4 >= (1ll << 32))
            return -2;
        GcnParams &g = sp.b[i].g;
        g = GcnParams{};
        g.x = a.xin; g.w = a.gcn_w; g.bias = a.gcn_bias; g.y = a.y_ring; g.ell_src = a.ell_src; g.ell_val = a.ell_val;
        for (int k = 0; k < 3; ++k) g.ell_cnt[k] = a.ell_cnt[k];
        g.ell_w = a.ell_w; g.adj_seg_stride = 0;
        g.x_seg_stride = (int64_t)a.c_in * P; g.x_chan_stride = P; g.y_seg_stride = (int64_t)a.c_out * P; g.y_chan_stride = P;
        g.Cin = a.c_in; g.CinPad = round_up(a.c_in, CSK_CPAD); g.Cout = a.c_out; g.Mpad = round_up(a.c_out, CSK_MT);
        g.frames = n_skel; g.V = V; g.R = a.gcn_res_mode == CSK_RES_CONV ? 4 : 3; g.res_mode = a.gcn_res_mode;
        g.vmagic = vmagic_of(V); g.mtiles = 1; g.qtiles = (unsigned)((Q + NP - 1) / NP); g.ksplit = 1; g.cper = g.CinPad; g.part = nullptr;
        g.x_ring_slots = a.xin_slots; g.x_ring_slot0 = a.xin_slot0; g.y_ring_slots = a.y_slots; g.y_ring_slot0 = a.y_slot0;
        g.stagger = stagger_units("CSK_GCN16_STAGGER", GCN16_STAGGER) | (prio_mode("CSK_GCN16_PRIO", GCN16_PRIO) << 20);
        g.stamps = nullptr;
        StepParams &t = sp.b[i].t;
        t = StepParams{};
        t.ring = a.y_ring; t.w = a.tcn_w; t.xres = a.xin; t.wres = nullptr; t.bias = a.tcn_bias; t.out = a.out;
        t.C = a.c_out; t.Cpad = round_up(a.c_out, CSK_CPAD); t.Cout = a.c_out; t.Mpad = round_up(a.c_out, CSK_MT);
        t.K = 9; t.slots = a.y_slots; t.head = a.y_slot0; t.head_step = 1; t.res_mode = a.res_mode;
        t.Cres = a.res_mode ? a.c_in : 1; t.CresPad = round_up(t.Cres, CSK_CPAD); t.relu = 1; t.P = P; t.fast_epi = 1;
        t.xres_slots = a.xin_slots; t.xres_slot0 = a.x_res_slot0; t.xres_step = 1; t.out_slots = a.out_slots; t.out_slot0 = a.out_slot0;
        t.stamps = nullptr; t.stagger = prio_mode("CSK_TCN16_PRIO", TCN16_PRIO) << 16;
        t.ksplit = 1; t.cper = t.Cpad; t.part = nullptr; t.gx = (unsigned)((P + NP - 1) / NP); t.gy = 1; t.gz = 1;
    }
    void (*kern)(CoStackParams) = best_nb == 25 ? co_stack16_kernel<25> : co_stack16_kernel<18>;
    const int NT = 16 * best_nb;
    // (identity-residual graph convs only: R = 3; their x tile is WinT16's: 10 floats per column)
    const size_t lds_g = (size_t)(8 * 3 * (80 + row16(NT)) + 10 * NT), lds_t = best_nb == 25 ? G16<25, 4, 1>::LDS_FLOATS : G16<18, 4, 1>::LDS_FLOATS;
    const size_t lds = (lds_g > lds_t ? lds_g : lds_t) * sizeof(float);
    if (const int e = csk_ensure_lds((const void *)kern, lds)) return e;
    hipLaunchKernelGGL(kern, dim3((unsigned)((P + NP - 1) / NP)), dim3(NTHREADS), lds, (hipStream_t)stream, sp);
    return (int)hipGetLastError();
}

extern "C" int csk_co_stack_step_f32(int n_blocks, const csk_co_block_args *b, int n_skel, int V, int64_t P, void *stream) {
    if (!b || n_blocks < 1 || n_blocks > CSK_CO_STACK_MAX) CSK_FAIL("co_stack_step: 1..%d blocks expected", CSK_CO_STACK_MAX);
    for (int i = 0; i + 1 < n_blocks; ++i)
        if (b[i + 1].xin != b[i].out || b[i + 1].xin_slots != b[i].out_slots || b[i + 1].xin_slot0 != b[i].out_slot0 ||
            b[i + 1].c_in != b[i].c_out)
            CSK_FAIL("co_stack_step: block %d does not read what block %d emits (ring, slot count, first slot, channels)", i + 1, i);
    bool fusable = true;                                      // what csk_co_block_step_f32 would reject must not reach the kernel
    for (int i = 0; i < n_blocks && fusable; ++i) {
        const csk_co_block_args &a = b[i];
        fusable = a.xin && a.gcn_w && a.gcn_bias && a.ell_src && a.ell_val && a.y_ring && a.tcn_w && a.tcn_bias && a.out && a.c_in > 0 &&
                  a.c_out > 0 && n_skel > 0 && V >= 2 && V <= 64 && P >= (int64_t)n_skel * V && !(P & 3) && P < (1ll << 31) - 256 &&
                  a.xin_slots >= 8 && a.y_slots >= 12 && a.out_slots >= 4 && a.xin_slot0 >= 0 && a.xin_slot0 < a.xin_slots &&
                  a.y_slot0 >= 0 && a.y_slot0 < a.y_slots && a.out_slot0 >= 0 && a.out_slot0 < a.out_slots && a.x_res_slot0 >= 0 &&
                  a.x_res_slot0 < a.xin_slots && (a.gcn_res_mode == CSK_RES_IDENTITY || a.gcn_res_mode == CSK_RES_CONV) &&
                  (a.gcn_res_mode != CSK_RES_IDENTITY || a.c_in == a.c_out) && (a.res_mode == CSK_RES_NONE || a.res_mode == CSK_RES_IDENTITY) &&
                  (a.res_mode != CSK_RES_IDENTITY || a.c_in == a.c_out) && a.ell_w >= 1 && a.ell_w <= V && a.ell_cnt[0] >= 0 &&
                  a.ell_cnt[1] >= 0 && a.ell_cnt[2] >= 0 && a.ell_cnt[2] <= a.ell_w &&
                  !(((uintptr_t)a.xin | (uintptr_t)a.y_ring | (uintptr_t)a.out) & 15);
    }
    if (fusable && n_blocks > 1) {
        const int rc = csk_launch_co_stack16(n_blocks, b, n_skel, V, P, stream);
        if (rc != -2) return rc;
    }
    for (int i = 0; i < n_blocks; ++i) {                      // per block (argument errors are reported from there)
        const csk_co_block_args &a = b[i];
        if (const int rc = csk_co_block_step_f32(a.xin, a.xin_slots, a.xin_slot0, a.c_in, a.gcn_w, a.gcn_bias, a.ell_src, a.ell_val, a.ell_cnt,
                                                 a.ell_w, a.gcn_res_mode, a.y_ring, a.y_slots, a.y_slot0, a.tcn_w, a.tcn_bias, a.res_mode,
                                                 a.x_res_slot0, a.out, a.out_slots, a.out_slot0, a.c_out, n_skel, V, P, stream))
            return rc;
    }
    return 0;
}
