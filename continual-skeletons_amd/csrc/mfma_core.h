// mfma_core.h -- device-side building blocks shared by the clip (tcn.hip, gcn.hip) and continual (step.hip) kernels:
// the fp32-MFMA "shifted GEMM" chunk, and the issue/commit staging helpers (register prefetch).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/cskel.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));   // native vector: stays in VGPRs (HIP float4 is a struct)

static constexpr int KC = CSK_KC;
static constexpr int NTHREADS = 256;

char *csk_err_buf();   // thread-local message buffer (defined in runtime.hip)
#define CSK_FAIL(...)                                   \
    do {                                                \
        snprintf(csk_err_buf(), 256, __VA_ARGS__);      \
        return -1;                                      \
    } while (0)

static inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

// Raise a kernel's dynamic-LDS cap once (and again only if a larger tile is requested): steady-state launches
// then consist of hipLaunchKernel alone.  Returns hipSuccess (0) or the error.
int csk_ensure_lds(const void *kernel, size_t bytes);
// diagnostic switches (runtime.hip): active only when CSK_DIAG was set when the library was loaded
bool csk_diag_flag(const char *name);
int csk_diag_int(const char *name);
unsigned long long *csk_diag_stamps();
static inline unsigned vmagic_of(int V) { return (unsigned)(((1ull << 32) + V - 1) / V); }

// ------------------------------------------------------------------------------------------------
// shared device pieces
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int div_magic(int x, unsigned magic) {
    // x / V for 0 <= x < 2^32 / V, magic = ceil(2^32 / V): one v_mul_hi_u32
    return (int)__umulhi((unsigned)x, magic);
}

// XCD-aware work-item id: workgroups are dealt round-robin over the 8 XCDs (b and b+8 share an L2), so a
// linear blockIdx puts neighbouring tiles -- which share temporal halos and weight panels -- on eight
// different L2s.  Remap so that every XCD walks a CONTIGUOUS range of work items (bijective for any grid
// size; placement is a speed assumption only, never a correctness one).
// Access through a wave-uniform row pointer plus a 32-bit per-lane BYTE offset: lets the compiler keep the row base in
// SGPRs (global_load/store "saddr" form) instead of materialising a 64-bit address per lane and access.
__device__ __forceinline__ float ld_lane(const float *row, unsigned byte_off) {
    return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(row) + byte_off);
}
__device__ __forceinline__ void st_lane(float *row, unsigned byte_off, float v) {
    *reinterpret_cast<float *>(reinterpret_cast<char *>(row) + byte_off) = v;
}

// ReLU that propagates NaN like torch.relu / np.maximum (v_maximum3_f32; fmaxf would return the non-NaN operand and
// mask corrupt activations as zeros)
__device__ __forceinline__ float relu_nan(float v) { return __builtin_elementwise_maximum(v, 0.f); }

__device__ __forceinline__ unsigned xcd_contiguous_id(unsigned bid, unsigned total) {
    const unsigned q = total / 8, r = total % 8, xcd = bid % 8, k = bid / 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// One K-chunk of the shifted GEMM for one wave: acc[mi][ni] += W[r][kk][rows] x B[r][kk][cols].
//   Wl : [taps][KC][MT]  (row = output channel contiguous -> A operand, lane i = l&31, k = l>>5)
//   Bl : B value of (tap r, channel kk, column) at Bl[r*tapB + kk*ldb + off_ni]
template <int MT>
__device__ __forceinline__ void mfma_chunk(const float *__restrict__ Wl, const float *__restrict__ Bl,
                                           int taps, int ldb, int tapB, int offA, int off0, int off1,
                                           int kh, f32x16 (&acc)[2][2]) {
    for (int r = 0; r < taps; ++r) {
        const float *wr = Wl + r * (KC * MT) + offA + kh * MT;
        const float *br = Bl + r * tapB + kh * ldb;
#pragma unroll
        for (int s = 0; s < KC / 2; ++s) {
            const float a0 = wr[2 * s * MT], a1 = wr[2 * s * MT + 32];
            const float b0 = br[2 * s * ldb + off0], b1 = br[2 * s * ldb + off1];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
}

// ---- staging, split into ISSUE (global -> registers) and COMMIT (registers -> LDS) so that the loads of
// chunk i+1 are in flight underneath the MFMAs of chunk i (register prefetch; the commit happens behind the
// barrier that retires chunk i's LDS reads).
// No predication anywhere (a predicated element makes hipcc branch around each load, wait vmcnt(0) per
// element and demote the array to scratch): out-of-range slots are CLAMPED to the last valid element, so
// surplus threads reload / rewrite the same value to the same address.  Per-thread offsets are chunk
// invariant and computed once; the per-chunk part of every address is wave-uniform (scalar base).
template <int MT>
struct WStage {
    static constexpr int M4 = MT / 4;
    static constexpr int WB = (9 * KC * M4 + NTHREADS - 1) / NTHREADS;   // f32x4 per thread for a 9-tap chunk
    unsigned goff[WB];   // element offset inside a chunk of packed weights (global [taps][Cpad][Mpad])
    unsigned loff[WB];   // element offset inside Wl [taps][KC][MT]
    f32x4 v[WB];
    __device__ __forceinline__ void setup(int taps, int Cpad, int Mpad, int tid) {
        const int last = taps * KC * M4 - 1;
#pragma unroll
        for (int u = 0; u < WB; ++u) {
            const int e = min(u * NTHREADS + tid, last);
            const int row = e / M4, m4 = e % M4;   // powers of two
            goff[u] = (unsigned)(((row / KC) * Cpad + (row % KC)) * Mpad + m4 * 4);
            loff[u] = (unsigned)(e * 4);
        }
    }
    __device__ __forceinline__ void issue_slot(int u, const float *__restrict__ chunk_base) {
        if (u < WB) v[u] = *reinterpret_cast<const f32x4 *>(chunk_base + goff[u]);
    }
    // chunk_base = w + c0 * Mpad + m0 (uniform)
    __device__ __forceinline__ void issue(const float *__restrict__ chunk_base) {
#pragma unroll
        for (int u = 0; u < WB; ++u) v[u] = *reinterpret_cast<const f32x4 *>(chunk_base + goff[u]);
    }
    __device__ __forceinline__ void commit(float *__restrict__ Wl) const {
#pragma unroll
        for (int u = 0; u < WB; ++u) *reinterpret_cast<f32x4 *>(Wl + loff[u]) = v[u];
    }
};

// WStage for a full 9-tap chunk of a 128-row tile: 9 * KC * 32 = 2304 f32x4 = exactly 9 per thread, and slot u of a thread
// is tap u, channel row tid / 32, rows 4 * (tid % 32) ..: the global and LDS offsets of the 9 slots differ by wave-uniform
// constants (Cpad * Mpad elements / KC * 128 floats), so ONE offset register each replaces the 18 of the general form.
struct WStage9x128 {
    unsigned goff0, loff0, gstride;
    f32x4 v[9];
    __device__ __forceinline__ void setup(int Cpad, int Mpad, int tid) {
        goff0 = (unsigned)((tid >> 5) * Mpad + (tid & 31) * 4);
        loff0 = (unsigned)(tid * 4);
        gstride = (unsigned)(Cpad * Mpad);
    }
    __device__ __forceinline__ void issue_slot(int u, const float *__restrict__ chunk_base) {
        v[u] = *reinterpret_cast<const f32x4 *>(chunk_base + (size_t)u * gstride + goff0);
    }
    __device__ __forceinline__ void issue(const float *__restrict__ chunk_base) {
#pragma unroll
        for (int u = 0; u < 9; ++u) issue_slot(u, chunk_base);
    }
    __device__ __forceinline__ void commit(float *__restrict__ Wl) const {
#pragma unroll
        for (int u = 0; u < 9; ++u) *reinterpret_cast<f32x4 *>(Wl + u * (KC * 128) + loff0) = v[u];
    }
};

// The same for a 64-row tile: 9 * KC * 16 = 1152 f32x4 = 4.5 per thread; slot u of a thread is tap 2 u + tid / 128, channel
// row (tid / 16) % 8; the upper half of the threads has no slot 4 and repeats its slot 3 (same value to the same address).
struct WStage9x64 {
    unsigned goff0, loff0, gstride, u4;
    f32x4 v[5];
    __device__ __forceinline__ void setup(int Cpad, int Mpad, int tid) {
        goff0 = (unsigned)(((tid >> 7) * Cpad + ((tid >> 4) & 7)) * Mpad + (tid & 15) * 4);
        loff0 = (unsigned)(tid * 4);
        gstride = (unsigned)(2 * Cpad * Mpad);
        u4 = tid >= 128 ? 3u : 4u;
    }
    __device__ __forceinline__ void issue_slot(int u, const float *__restrict__ chunk_base) {
        if (u < 4) v[u] = *reinterpret_cast<const f32x4 *>(chunk_base + (size_t)u * gstride + goff0);
        else if (u == 4) v[4] = *reinterpret_cast<const f32x4 *>(chunk_base + u4 * gstride + goff0);
    }
    __device__ __forceinline__ void issue(const float *__restrict__ chunk_base) {
#pragma unroll
        for (int u = 0; u < 5; ++u) issue_slot(u, chunk_base);
    }
    __device__ __forceinline__ void commit(float *__restrict__ Wl) const {
#pragma unroll
        for (int u = 0; u < 4; ++u) *reinterpret_cast<f32x4 *>(Wl + u * 1024 + loff0) = v[u];
        *reinterpret_cast<f32x4 *>(Wl + u4 * 1024 + loff0) = v[4];
    }
};

// activations: KC channel rows x span positions of one segment -> Bl [KC][ldb]; positions outside [0, TV)
// and channels >= C read as zero (conv zero padding / channel padding): the address is clamped into the
// tensor and the value replaced by 0 with a select, so the load itself is unconditional.
template <int NJ, int RPW_ = KC / (NTHREADS / 64)>
struct BStage {
    static constexpr int RPW = RPW_;                   // rows per wave (2 for an 8-channel chunk; 2 G for G channel groups)
    // slot u of a lane is position j = min(u * 64 + lane, span - 1) of the LDS row = position clamp(pbase + j) of the channel
    // row; both are recomputed where they are used (two or three integer operations per access, boundary tiles only) instead
    // of 2 NJ registers held across the K loop -- the 9-tap kernels run at the 256-register limit
    int lane_, spanm1, pbase_, tvm1;
    unsigned valid;      // bit u: position is inside [0, TV)
    float v[RPW][NJ];
    __device__ __forceinline__ void setup(int pbase, int span, int TV, int lane) {
        valid = 0;
        lane_ = lane;
        spanm1 = span - 1;
        pbase_ = pbase;
        tvm1 = TV - 1;
#pragma unroll
        for (int u = 0; u < NJ; ++u) {
            const int pp = pbase + min(u * 64 + lane, span - 1);
            valid |= (pp >= 0 && pp < TV) ? (1u << u) : 0u;
        }
    }
    // (the empty asm keeps the compiler from hoisting the NJ offsets out of the K loop into registers again)
    __device__ __forceinline__ unsigned goff(int u) const {
        int l = lane_;
        asm volatile("" : "+v"(l));
        return (unsigned)min(max(pbase_ + min(u * 64 + l, spanm1), 0), tvm1);
    }
    // slot u of every row this wave owns (used to trickle the loads between MFMA groups)
    __device__ __forceinline__ void issue_slot(int u, const float *__restrict__ seg_base, int C, int64_t chan_stride,
                                               int c0, int wave) {
        if (u >= NJ) return;
        const unsigned g = goff(u);
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            const int c = c0 + wave + rr * (NTHREADS / 64);
            const float *src = seg_base + (int64_t)min(c, C - 1) * chan_stride;
            const float x = src[g];
            v[rr][u] = ((c < C ? valid : 0u) >> u) & 1u ? x : 0.f;
        }
    }
    // wave is wave-uniform (readfirstlane); rows c0 + wave + 4*rr
    __device__ __forceinline__ void issue(const float *__restrict__ seg_base, int C, int64_t chan_stride, int c0,
                                          int wave) {
#pragma unroll
        for (int u = 0; u < NJ; ++u) issue_slot(u, seg_base, C, chan_stride, c0, wave);
    }
    // third G of the next chunk's loads: sweeps 3G .. 3G+2 (plus 3G+9 .. for long spans)
    template <int G>
    __device__ __forceinline__ void issue_third(const float *__restrict__ seg_base, int C, int64_t chan_stride, int c0,
                                                int wave) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            issue_slot(3 * G + j, seg_base, C, chan_stride, c0, wave);
            if (NJ > 9) issue_slot(3 * G + j + 9, seg_base, C, chan_stride, c0, wave);
        }
    }
    __device__ __forceinline__ void commit(float *__restrict__ Bl, int ldb, int wave) const {
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            float *dst = Bl + (wave + rr * (NTHREADS / 64)) * ldb;
#pragma unroll
            for (int u = 0; u < NJ; ++u) {
                int l = lane_;
                asm volatile("" : "+v"(l));
                dst[min(u * 64 + l, spanm1)] = v[rr][u];
            }
        }
    }
};

// Vector form of BStage for tiles whose whole staged span lies inside the activation row (no conv zero padding to
// apply): 16-byte global loads (4-byte aligned -- rows and tap shifts are not 16-byte aligned in general; the LDS side is)
// and ds_write_b128, i.e. a quarter of the load and LDS-write instructions of the scalar form.
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
template <int NJ4>
struct BStage4 {
    static_assert(NJ4 <= 4, "issue_third covers sweeps 0..3 only: a fifth sweep would never be loaded (stale LDS rows)");
    static constexpr int RPW = KC / (NTHREADS / 64);   // rows per wave
    unsigned goff[NJ4];   // first position of the vector inside a channel row
    unsigned loff[NJ4];   // ... inside an LDS row
    f32x4 v[RPW][NJ4];
    __device__ __forceinline__ void setup(int pbase, int span, int lane) {
        const int nvec = (span + 3) / 4;
#pragma unroll
        for (int u = 0; u < NJ4; ++u) {
            const int i = min(u * 64 + lane, nvec - 1);
            goff[u] = (unsigned)(pbase + 4 * i);
            loff[u] = (unsigned)(4 * i);
        }
    }
    __device__ __forceinline__ void issue_slot(int u, const float *__restrict__ seg_base, int C, int64_t chan_stride,
                                               int c0, int wave) {
        if (u >= NJ4) return;
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            const int c = min(c0 + wave + rr * (NTHREADS / 64), C - 1);     // clamped: padding channels carry zero weights
            v[rr][u] = *reinterpret_cast<const f32x4u *>(seg_base + (int64_t)c * chan_stride + goff[u]);
        }
    }
    __device__ __forceinline__ void issue(const float *__restrict__ seg_base, int C, int64_t chan_stride, int c0, int wave) {
#pragma unroll
        for (int u = 0; u < NJ4; ++u) issue_slot(u, seg_base, C, chan_stride, c0, wave);
    }
    // third G of the next chunk's loads: sweep G (and sweep 3 with the first third)
    template <int G>
    __device__ __forceinline__ void issue_third(const float *__restrict__ seg_base, int C, int64_t chan_stride, int c0,
                                                int wave) {
        issue_slot(G, seg_base, C, chan_stride, c0, wave);
        if (G == 0) issue_slot(3, seg_base, C, chan_stride, c0, wave);
    }
    __device__ __forceinline__ void commit(float *__restrict__ Bl, int ldb, int wave) const {
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            float *dst = Bl + (wave + rr * (NTHREADS / 64)) * ldb;
#pragma unroll
            for (int u = 0; u < NJ4; ++u) *reinterpret_cast<f32x4 *>(dst + loff[u]) = v[rr][u];
        }
    }
};

// Taps [r0, r1) of a chunk (rolled loop; same body as mfma_chunk).  The TCN kernels run a 9-tap chunk as three
// 3-tap segments with one third of the NEXT chunk's global loads issued in front of each, instead of one burst of
// 27 loads: measured with in-kernel stamps, the burst cost every wave 2.1-3.7 k cycles per chunk in VMEM-issue
// stalls (all 8 waves of a CU hit the address unit at once and an in-order wave cannot issue MFMAs meanwhile).
template <int MT>
__device__ __forceinline__ void mfma_taps(const float *__restrict__ Wl, const float *__restrict__ Bl, int r0, int r1,
                                          int ldb, int tapB, int offA, int off0, int off1, int kh,
                                          f32x16 (&acc)[2][2]) {
    // Plain loop, scheduled by the compiler: tools/microbench/mfma_loop_probe.hip measures this form at 152.6 TFLOP/s
    // (97 % of the fp32-MFMA peak) with LDS operands at 2 waves/SIMD, against 145 for a hand-pipelined variant
    // fenced with sched_barrier(0) and 153 with register operands -- the LDS fragment reads are free here.
    for (int r = r0; r < r1; ++r) {
        const float *wr = Wl + r * (KC * MT) + offA + kh * MT;
        const float *br = Bl + r * tapB + kh * ldb;
#pragma unroll
        for (int s = 0; s < KC / 2; ++s) {
            const float a0 = wr[2 * s * MT], a1 = wr[2 * s * MT + 32];
            const float b0 = br[2 * s * ldb + off0], b1 = br[2 * s * ldb + off1];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
}

// Compile-time form of mfma_taps for NTAPS consecutive taps starting at r0: straight-line code, so the compiler can issue
// the LDS fragment reads of tap r + 1 under the MFMAs of tap r.  The rolled loop above exposes the LDS latency once per
// tap, which a wave only hides while its SIMD partner (the other workgroup's wave) is issuing MFMAs too -- not while the
// partner sits in its barrier / commit phase and this wave has the matrix pipe to itself.
// AH > 0: the operand reads of k-step i + AH are issued under the MFMAs of k-step i (sched_group_barrier; see mfma16_read_ahead
// in tile16.h).  For launches whose workgroups run ALONE on their CU (the split-K clip latency path: one wave per SIMD, nobody
// to fill the LDS round trip the default schedule -- read, s_waitcnt lgkmcnt(0), four MFMAs -- exposes per k-step); with two
// workgroups per CU and many rounds it measured nothing (clip forward, batch 256: 69.7-70.1 ms either way).
template <int MT, int NTAPS, int AH = 0>
__device__ __forceinline__ void mfma_taps_ct(const float *__restrict__ Wl, const float *__restrict__ Bl, int r0, int ldb, int tapB,
                                             int offA, int off0, int off1, int kh, f32x16 (&acc)[2][2]) {
    const float *wr = Wl + r0 * (KC * MT) + offA + kh * MT;
    const float *br = Bl + r0 * tapB + kh * ldb;
#pragma unroll
    for (int r = 0; r < NTAPS; ++r) {
#pragma unroll
        for (int s = 0; s < KC / 2; ++s) {
            const float a0 = wr[r * (KC * MT) + 2 * s * MT], a1 = wr[r * (KC * MT) + 2 * s * MT + 32];
            const float b0 = br[r * tapB + 2 * s * ldb + off0], b1 = br[r * tapB + 2 * s * ldb + off1];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
    if (AH > 0) {                                                  // per k-step: one ds_read2 (A) + two ds_read (B), four MFMAs
        __builtin_amdgcn_sched_group_barrier(0x100, 3 * AH, 0);
#pragma unroll
        for (int i = 0; i < NTAPS * (KC / 2); ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
        }
    }
}
