// tile16.h -- device-side building blocks of the 16x16x4 tile family (v_mfma_f32_16x16x4_f32; wave tile = 16 output channels x
// NB column blocks of 16): register-prefetch staging, the per-tap MFMA sweep and the 16-byte epilogue.  Shared by the continual
// step kernels (step16.hip) and the clip temporal conv (tcn16.hip).
#pragma once
#include "mfma_core.h"

namespace {


// Time model of a launch: the chip is MFMA-bound, so a CU delivers the same work per unit time with one or two workgroups
// resident; what counts is the largest number of tiles ONE CU has to run (tiles dealt evenly over the 256 CUs of an MI355X)
// times the columns a 64-row tile carries (padding included).
constexpr int64_t CUS = 256;
// start stagger of the odd-slot workgroup in units of 64 cycles (stagger_odd_slot); CSK_*_STAGGER under CSK_DIAG=1 overrides
// (value + 1: 1 = off)
constexpr int GCN16_STAGGER = 0, TCN16_STAGGER = 0;       // swept in round 6: no effect (+- 0.1 %)
inline int stagger_units(const char *env, int dflt) {
    const int v = csk_diag_int(env);
    return v > 0 ? v - 1 : dflt;
}
// which workgroup of a CU the issue arbiter favours inside the MFMA segments (step16.hip, tcn16_tile / gcn16_tile): 0 = equal
// priorities (= the older one, always), 1 = alternating chunk by chunk, 2 = always the younger one; CSK_*_PRIO under CSK_DIAG=1
// overrides (value + 1).  Round 6, 1024 NTU streams, same process: 985 -> 1 004 k frames/s (temporal step) -> 1 011 k (+ graph conv).
constexpr int TCN16_PRIO = 1, GCN16_PRIO = 1;
inline int prio_mode(const char *env, int dflt) {
    const int v = csk_diag_int(env);
    return v > 0 ? v - 1 : dflt;
}
inline double cost_model(int64_t tiles64, double tile_cols) { return (double)((tiles64 + CUS - 1) / CUS) * tile_cols; }

constexpr int imax(int a, int b) { return a > b ? a : b; }
constexpr int row16(int n) { return ((n - 16 + 31) / 32) * 32 + 16; }   // smallest stride >= n that is 16 (mod 32)

// Staging of KCHX channel rows x NSL ring slots x NP positions (register prefetch: issue = global -> registers, commit =
// registers -> LDS).  Unit e of the (channel, slot, quad) space, quad fastest; a thread owns units sweep * 256 + tid.
template <int KCHX, int NSL, int NP, int ROWX, bool EVENODD, bool PARTIAL = false>
struct Win16 {
    static constexpr int Q = NP / 4, U = KCHX * NSL * Q, NSW = (U + NTHREADS - 1) / NTHREADS, NEV = (NSL + 1) / 2;
    static_assert(KCHX * ROWX * 4 < 65536, "LDS byte offsets are packed two to a register");
    unsigned goff[NSW];            // byte offset from (ring + c0 * P + p0): the ring is < 4 GB (checked by the launcher)
    unsigned loff2[(NSW + 1) / 2]; // LDS byte offsets of sweeps 2 i (low half) and 2 i + 1 (high half)
    unsigned gback[PARTIAL ? NSW : 1];   // PARTIAL: bytes to step back in the chunk that holds only `nreal` real channel rows (rows
                                         // past them re-read the last real row: their weights are zero, the value only has to be finite)
    f32x4 v[NSW];
    // slot w of the window = slot (first + w * step) % slots of the source (slot_stride floats apart, channel rows chan_stride
    // apart); pmax = last legal f32x4 start relative to the tile's first position
    __device__ __forceinline__ void setup(int first, int step, int slots, int64_t slot_stride, int64_t chan_stride, int pmax, int tid,
                                          int nreal = KCHX) {
#pragma unroll
        for (int u = 0; u < NSW; ++u) {
            const int e = min(u * NTHREADS + tid, U - 1);
            const int i = e % Q, rw = e / Q, w = rw % NSL, kk = rw / NSL;
            const int64_t so = (int64_t)((first + w * step) % slots) * slot_stride + (int64_t)kk * chan_stride + min(4 * i, pmax);
            goff[u] = (unsigned)(so * 4);
            if (PARTIAL) gback[u] = (unsigned)((int64_t)max(kk - (nreal - 1), 0) * chan_stride * 4);
            const int sl = EVENODD ? ((w & 1) ? (NEV + (w >> 1)) * NP : (w >> 1) * NP) : w * NP;
            const unsigned lo = (unsigned)(kk * ROWX + sl + 4 * i) * 4u;
            if (u & 1) loff2[u / 2] |= lo << 16;
            else loff2[u / 2] = lo;
        }
    }
    // The load address is a wave-uniform base (scalar registers) + one 32-bit lane offset; the empty asm keeps the compiler
    // from folding the loop-invariant part of the base into a 64-bit per-lane address held across the K loop.
    template <int U0, int U1>
    __device__ __forceinline__ void issue_range(const float *__restrict__ base) {
#pragma unroll
        for (int u = U0; u < U1; ++u) {
            unsigned g = goff[u];
            asm volatile("" : "+v"(g));
            v[u] = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(base) + g);
        }
    }
    template <int G>
    __device__ __forceinline__ void issue_third(const float *__restrict__ base) { issue_range<G * NSW / 3, (G + 1) * NSW / 3>(base); }
    __device__ __forceinline__ void issue(const float *__restrict__ base) { issue_range<0, NSW>(base); }
    // PARTIAL: `partial` (wave-uniform) picks the clamped offsets
    __device__ __forceinline__ void issue_sel(const float *__restrict__ base, bool partial) {
#pragma unroll
        for (int u = 0; u < NSW; ++u) {
            unsigned g = goff[u] - (partial ? gback[PARTIAL ? u : 0] : 0u);
            asm volatile("" : "+v"(g));
            v[u] = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(base) + g);
        }
    }
    // chunk whose channels c0 .. c0 + KCHX - 1 reach past C: rows >= C are read from row C - 1 and zeroed (their weights are
    // zero as well; the product must not be 0 x Inf)
    __device__ __forceinline__ void issue_tail(const float *__restrict__ base, int c0, int C, int64_t chan_stride, int tid) {
#pragma unroll
        for (int u = 0; u < NSW; ++u) {
            const int e = min(u * NTHREADS + tid, U - 1);
            const int kk = (e / Q) / NSL;
            const int over = max(c0 + kk - (C - 1), 0);
            const f32x4 x = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(base) + goff[u] - (size_t)over * chan_stride * 4);
            v[u] = x * (over ? 0.f : 1.f);
        }
    }
    __device__ __forceinline__ void commit(float *__restrict__ Bl) const {
#pragma unroll
        for (int u = 0; u < NSW; ++u) {
            const unsigned lo = (u & 1) ? loff2[u / 2] >> 16 : loff2[u / 2] & 0xffffu;
            *reinterpret_cast<f32x4 *>(reinterpret_cast<char *>(Bl) + lo) = v[u];
        }
    }
};

// Channel-INTERLEAVED staging of 8 channel rows x NSL slots x NP positions for the graph conv's aggregation phase: position
// `pos` of the tile (slot-major, 0 .. NSL * NP - 1) keeps channels 0-3 in four consecutive floats at xt_off(pos) and channels
// 4-7 at XT_HALF + xt_off(pos), so a lane fetches the 8 channels of one source joint with two ds_read_b128 instead of eight
// ds_read_b32 (the aggregation phase is bound by the NUMBER of LDS instructions: 48 gathers per column and chunk before).
// A position quad is 16 floats + 4 of padding: consecutive positions stay 16 bytes apart inside a quad (the gathers of
// neighbouring columns fall on neighbouring banks) and the transposing commit -- lanes = (quad, channel), four ds_write_b32
// each -- spreads 8 quads x 4 channels over all 32 banks.  Unit e = (slot, quad, channel), CHANNEL fastest: a wave's load
// covers 8 quads = 128 contiguous bytes of each of the 8 channel rows.
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int XT_QUAD = 20;
__host__ __device__ constexpr int xt_off(int pos) { return (pos >> 2) * XT_QUAD + (pos & 3) * 4; }
template <int NSL, int NP, bool PARTIAL = false, int NTH = NTHREADS>
struct WinT16 {
    static constexpr int Q = NP / 4, U = 8 * NSL * Q, NSW = (U + NTH - 1) / NTH, HALF = NSL * Q * XT_QUAD;
    static_assert(2 * HALF * 4 < 65536 * 4, "LDS offsets");
    unsigned goff[NSW];
    unsigned short loff[NSW];      // float offset of (quad, channel) in the interleaved tile
    unsigned gback[PARTIAL ? NSW : 1];
    f32x4 v[NSW];
    __device__ __forceinline__ void setup(int first, int step, int slots, int64_t slot_stride, int64_t chan_stride, int pmax, int tid,
                                          int nreal = 8) {
#pragma unroll
        for (int u = 0; u < NSW; ++u) {
            const int e = min(u * NTH + tid, U - 1);
            const int kk = e & 7, qi = e >> 3, i = qi % Q, w = qi / Q;
            const int64_t so = (int64_t)((first + w * step) % slots) * slot_stride + (int64_t)kk * chan_stride + min(4 * i, pmax);
            goff[u] = (unsigned)(so * 4);
            if (PARTIAL) gback[u] = (unsigned)((int64_t)max(kk - (nreal - 1), 0) * chan_stride * 4);
            loff[u] = (unsigned short)((kk >> 2) * HALF + qi * XT_QUAD + (kk & 3));
        }
    }
    __device__ __forceinline__ void issue_sel(const float *__restrict__ base, bool partial) {
#pragma unroll
        for (int u = 0; u < NSW; ++u) {
            unsigned g = goff[u] - (partial ? gback[PARTIAL ? u : 0] : 0u);
            asm volatile("" : "+v"(g));
            v[u] = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(base) + g);
        }
    }
    __device__ __forceinline__ void commit(float *__restrict__ Xt) const {
#pragma unroll
        for (int u = 0; u < NSW; ++u) {
            float *d = Xt + loff[u];
#pragma unroll
            for (int j = 0; j < 4; ++j) d[4 * j] = v[u][j];
        }
    }
};

// weights: NTAPS x KCHX channel rows x 64 output channels of the packed [tap][Cpad][Mpad] layout -> Wl[(r * KCHX + kk)][LDW];
// ENTRY: rows in the graph conv's k order instead (gcn_entry below)
__host__ __device__ constexpr int gcn_entry(int kk, int r, int R) { return ((kk >> 1) * R + r) * 2 + (kk & 1); }
template <int NTAPS, int KCHX, int LDW, bool ENTRY = false, int NTH = NTHREADS>
struct W16 {
    static constexpr int U = NTAPS * KCHX * 16, NSW = (U + NTH - 1) / NTH;
    unsigned goff[NSW], loff[NSW];
    f32x4 v[NSW];
    __device__ __forceinline__ void setup(int Cpad, int Mpad, int tid) {
#pragma unroll
        for (int u = 0; u < NSW; ++u) {
            const int e = min(u * NTH + tid, U - 1);
            const int row = e / 16, m4 = e % 16, r = row / KCHX, kk = row % KCHX;
            goff[u] = (unsigned)(((r * Cpad + kk) * Mpad + m4 * 4) * 4);
            loff[u] = (unsigned)((ENTRY ? gcn_entry(kk, r, NTAPS) : row) * LDW + m4 * 4);
        }
    }
    __device__ __forceinline__ void issue_one(int u, const float *__restrict__ base) {
        if (u < NSW) {
            unsigned g = goff[u];
            asm volatile("" : "+v"(g));
            v[u] = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(base) + g);
        }
    }
    __device__ __forceinline__ void issue(const float *__restrict__ base) {
#pragma unroll
        for (int u = 0; u < NSW; ++u) issue_one(u, base);
    }
    __device__ __forceinline__ void commit(float *__restrict__ Wl) const {
#pragma unroll
        for (int u = 0; u < NSW; ++u) *reinterpret_cast<f32x4 *>(Wl + loff[u]) = v[u];
    }
};

// The two workgroups of a CU start together and do identical work: left alone they run in LOCKSTEP -- both in their
// matrix-free phase (graph conv: aggregation; temporal step: commit + barriers) at the same time, the matrix pipe idle, then
// both contending for it.  The workgroup in the odd wave slot of its SIMDs (HW_ID.wave_id) therefore starts `units` x 64
// cycles late: one phase behind its partner, where it stays (matrix beside memory).  Speed only, never correctness.
__device__ __forceinline__ void stagger_odd_slot(int units) {
    if (units > 0 && (__builtin_amdgcn_s_getreg(6148) & 1))                     // HW_REG_HW_ID bits [3:0]: wave slot on the SIMD
        for (int i = 0; i < units; ++i) __builtin_amdgcn_s_sleep(1);
}

// Operand reads AHEAD of their use.  Left to itself the scheduler emits `ds_read2_b32; s_waitcnt lgkmcnt(0); mfma; mfma` 13
// times per tap: every MFMA pair waits for the LDS round trip of the read issued right in front of it, and a wave that has
// the matrix pipe to itself (its SIMD partner in a barrier / commit phase, or gone) runs at ~56 cycles per 32-cycle MFMA.
// The group barriers below order a straight-line run of NM MFMAs and their reads as: 1 + CSK_READ_AHEAD reads, then (two
// MFMAs, one read) over and over -- the read of pair i + CSK_READ_AHEAD is in flight under pair i.  Round 6, 1024 NTU
// streams, same box: 1 009 k -> 1 035 k frames/s at 2 (1: 997 k; 3: 1 034 k; one schedule per 3-tap segment instead of per tap:
// 1 021 k).  The clip kernels (many rounds per launch, the partner rarely away) measured no change with the same hints.
#ifndef CSK_READ_AHEAD
#define CSK_READ_AHEAD 2
#endif
template <int NM, int AH = CSK_READ_AHEAD>
__device__ __forceinline__ void mfma16_read_ahead() {
    if (AH > 0) {
        __builtin_amdgcn_sched_group_barrier(0x100, 1 + AH, 0);           // DS reads: the weight fragment + the first activation pairs
#pragma unroll
        for (int i = 0; i < (NM + 1) / 2; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);            // two MFMAs
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);            // one DS read
        }
    }
}

// one tap (one k-step of 4 channels): acc[cb] += act[16 cb .. + 15][k] x w[k][16 channels]
template <int NB, int AH = CSK_READ_AHEAD>
__device__ __forceinline__ void mfma16_tap(const float *__restrict__ wl, const float *__restrict__ bl, f32x4 (&acc)[NB]) {
    const float wf = wl[0];
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(bl[16 * cb], wf, acc[cb], 0, 0, 0);
    mfma16_read_ahead<NB, AH>();
}
// ... for a wave whose LAST column block may lie past the tile (`last` wave-uniform: the half-tile waves of an odd block count)
template <int NB, int AH = CSK_READ_AHEAD>
__device__ __forceinline__ void mfma16_tap_opt_last(const float *__restrict__ wl, const float *__restrict__ bl, f32x4 (&acc)[NB], bool last) {
    const float wf = wl[0];
#pragma unroll
    for (int cb = 0; cb < NB - 1; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(bl[16 * cb], wf, acc[cb], 0, 0, 0);
    mfma16_read_ahead<NB - 1, AH>();
    if (last) acc[NB - 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(bl[16 * (NB - 1)], wf, acc[NB - 1], 0, 0, 0);
}

// Epilogue shared by the kernels of this file: out = [ReLU](acc + bias + identity residual).  A lane holds positions
// 16 cb + 4 kq .. + 3 (columns = NSLOT slots x NP positions, slot-major) of output channel `ch` for every column block: 16-byte
// residual loads and stores.  Slot bases are wave-uniform byte offsets (the rings are < 4 GB); the slot of a column block is a
// compile-time constant except for the blocks that straddle a slot boundary.  Column blocks go in groups of EG: the group's
// residual loads are in flight together.  nrow = positions of the tile that lie inside the channel row (a multiple of 4: what
// may be loaded), nval = positions that are stored (<= nrow; a quad that straddles nval is stored element by element).
// UNAL: channel rows that are not whole 16-byte quads (clip tensors with T * V not a multiple of 4): 4-byte-aligned vector
// accesses, and the quad that straddles the end of a row is loaded element by element (never clamped: its values would shift).
typedef float f32x4u16 __attribute__((ext_vector_type(4), aligned(4)));
// c_off: first column of the wave's column blocks (a wave that owns only part of the tile's blocks; columns past the tile are
// masked like the ragged end of a tile).
template <int NB, int NSLOT, int NP, bool UNAL = false, int EG = 5>
__device__ __forceinline__ void epilogue16(f32x4 (&acc)[NB], const float *__restrict__ bias_p, int Cout, int ch, int kq, bool ident, bool relu,
                                           const float *__restrict__ xres, float *__restrict__ out, const unsigned (&xslot)[NSLOT],
                                           const unsigned (&oslot)[NSLOT], int64_t x_chan_stride, int64_t o_chan_stride, int p0, int nrow,
                                           int nval, int c_off = 0) {
    const bool chv = ch < Cout;
    const int chc = min(ch, Cout - 1);
    const float bias = bias_p[chc];
    const unsigned xrow = (unsigned)(((int64_t)chc * x_chan_stride + p0) * 4), orow = (unsigned)(((int64_t)chc * o_chan_stride + p0) * 4);
    const int pmax = nrow - 4;
    constexpr int NG = (NB + EG - 1) / EG;
    // group g + 1's residual loads are issued in front of group g's arithmetic and stores (two groups of registers)
    f32x4 rv[2][EG];
    unsigned oo[2][EG];
    int left[2][EG];                                         // stored positions from the quad's first on (<= 0: none)
    auto load_group = [&](int g, int b) {
#pragma unroll
        for (int u = 0; u < EG; ++u) {
            const int cb = g * EG + u;
            if (cb >= NB) continue;
            const int c = 16 * cb + 4 * kq + c_off;          // first column of this lane's quad: slot j, position pp
            int j = 0;
#pragma unroll
            for (int jj = 1; jj < NSLOT; ++jj) j += (c >= jj * NP) ? 1 : 0;
            const int pp = c - j * NP;
            unsigned os = oslot[0], xs = xslot[0];
#pragma unroll
            for (int jj = 1; jj < NSLOT; ++jj) { os = (j == jj) ? oslot[jj] : os; xs = (j == jj) ? xslot[jj] : xs; }
            left[b][u] = chv ? nval - pp : 0;
            if constexpr (UNAL) {
                const unsigned po = 4u * (unsigned)pp;
                oo[b][u] = os + orow + po;
                const float *src = reinterpret_cast<const float *>(reinterpret_cast<const char *>(xres) + (xs + xrow + po));
                f32x4 r = {0.f, 0.f, 0.f, 0.f};
                if (ident) {
                    if (left[b][u] >= 4) r = *reinterpret_cast<const f32x4u16 *>(src);
                    else {
#pragma unroll
                        for (int q = 0; q < 3; ++q)
                            if (q < left[b][u]) r[q] = src[q];
                    }
                }
                rv[b][u] = r;
            } else {
                const unsigned po = 4u * (unsigned)max(min(pp, pmax), 0);
                oo[b][u] = os + orow + po;
                rv[b][u] = ident ? *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(xres) + (xs + xrow + po)) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    load_group(0, 0);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int b = g & 1;
        if (g + 1 < NG) load_group(g + 1, b ^ 1);
#pragma unroll
        for (int u = 0; u < EG; ++u) {
            const int cb = g * EG + u;
            if (cb >= NB) continue;
            f32x4 o;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float v = acc[cb][q] + bias + rv[b][u][q];
                o[q] = relu ? relu_nan(v) : v;
            }
            float *dst = reinterpret_cast<float *>(reinterpret_cast<char *>(out) + oo[b][u]);
            if (left[b][u] >= 4) {
                if constexpr (UNAL) *reinterpret_cast<f32x4u16 *>(dst) = o;
                else *reinterpret_cast<f32x4 *>(dst) = o;
            } else if (left[b][u] > 0) {                     // the quad straddles the last stored position
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    if (q < left[b][u]) dst[q] = o[q];
            }
        }
    }
}

}  // namespace
