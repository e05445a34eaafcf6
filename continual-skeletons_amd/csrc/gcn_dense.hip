// gcn_dense.hip -- GCN stage for a DENSE adjacency that differs per sample (A-GCN clip form, models/a_gcn/a_gcn.py:48-69) or
// per skeleton frame (CoAGCN step form, models/coa_gcn/coa_gcn.py:11-14):
//   y = ReLU( sum_k W'_k . (x . adj_k) + b' + gcn_residual(x) ),   adj_k = softmax(...) + (A + graph_attn)_k, V x V, dense.
// The aggregated B operand of the channel-mixing GEMM is formed ON THE FLY, as in gcn_stage_sparse2_kernel, but from the
// whole adjacency column: a lane owns ONE output column (frame f, joint w) for the whole kernel and keeps the 3 V values
// adj_k[f][:, w] in registers (54 VGPRs at V = 18); per MFMA k-step it reads the V values x[c][f, :] of its channel from the
// LDS tile (five ds_read_b128, the same addresses for every lane of a frame: broadcast) and forms
//   B_k[c][f, w] = sum_v x[c][f, v] adj_k[f][v, w]          27 v_pk_fma_f32 (even / odd v in the two halves) + 3 adds
// in front of the 12-16 MFMAs that consume them.  No aggregated tile in LDS, no aggregation phase, one barrier per 8-channel
// chunk (ping-pong LDS, register prefetch two chunks ahead).  So that one set of adjacency registers serves all the MFMAs
// of a k-step, a wave's tile is MT rows x 32 columns (four / two accumulator blocks stacked over ONE column block): the
// vector-ALU work is 27 packed FMAs per 12 (MT = 128) or 6 (MT = 64) MFMAs, i.e. 14 % / 28 % of the matrix pipe's cycles,
// issued in its shadow.  Accumulator block i, MFMA row rho is output row NB rho + i of the tile: the A operands of the NB
// blocks are then ONE 16- / 8-byte LDS read of the row-major weight panel.
// Tiles are frame-aligned (FT = 128 / V whole frames, 126 of 128 columns at V = 18); x rows are staged with every frame
// padded to a multiple of 4 positions so that the lane's reads are aligned 16-byte vectors.
// Per-segment and per-frame adjacencies differ only in the address the prologue loads a lane's column from.
// Summation order over v: even and odd joints in separate fmaf chains, then one add (the general kernel of gcn.hip sums
// v = 0, 1, 2, ...): same tolerance class as every other fp32 path here, NOT bitwise equal to gcn_stage_kernel.
// Built for V = 18 (Kinetics / OpenPose-18, BASELINE configs[3]) and V = 25 (NTU RGB+D: 75 adjacency registers, joints staged
// one by one because frames start at odd offsets); other joint counts use gcn.hip.
#include <type_traits>

#include "mfma_core.h"
#include "gcn_params.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MT, bool CONVRES, int V, int KC2>
__global__ __launch_bounds__(NTHREADS, 2) void gcn_stage_dense2_kernel(const GcnParams p) {
    constexpr int VP = (V + 1) / 2, VPAD = (V + 3) & ~3; // joint pairs (odd V: the last pair is half empty), padded frame length in LDS
    constexpr bool EVEN = V % 2 == 0;
    constexpr int NT = 128, FT = NT / V;                 // tile columns (4 waves x 32), whole frames per tile
    constexpr int NB = MT / 32;                          // accumulator blocks of a wave
    constexpr int R = CONVRES ? 4 : 3;
    constexpr int NS = KC2 / 2, NH = NS / 2;             // (KC2 channels per chunk) MFMA k-steps per chunk, ... carrying loads
    constexpr int LDX = FT * VPAD;                       // x row length in LDS
    constexpr int M4 = MT / 4;
    constexpr int WB = (R * KC2 * M4 + NTHREADS - 1) / NTHREADS;   // f32x4 of weights per thread and chunk
    constexpr int XSL = EVEN ? 64 : 128;                 // staging slots per activation row: joint pairs (even V) or joints
    constexpr int XB = KC2 * XSL / NTHREADS;             // ... per thread and chunk
    constexpr int NL = WB + XB;
    static_assert(FT * V <= NT && VPAD % 4 == 0 && V <= 26 && KC2 * XSL % NTHREADS == 0, "tile shape");
    constexpr int WSZ = R * KC2 * MT, BUFSZ = WSZ + KC2 * LDX;      // one chunk buffer: Wl [R][KC2][MT], Bx [KC2][LDX]
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5;
    const unsigned wid = xcd_contiguous_id(blockIdx.x, gridDim.x);
    const int m0 = (int)(wid % p.mtiles) * MT, qt = (int)((wid / p.mtiles) % p.qtiles);
    const int seg = (int)(wid / (p.mtiles * p.qtiles));
    const int Q = p.frames * V;
    const int ta = qt * FT, q0 = ta * V;
    const int fcnt = min(FT, p.frames - ta);             // frames of this tile
    const int ncol = fcnt * V;

    // this lane's output column and its adjacency columns (registers for the whole kernel)
    const int j = wave * 32 + l31;
    const bool jv = j < ncol;
    const int jf = jv ? j / V : 0, jw = jv ? j - jf * V : 0;
    const int xoff = jf * VPAD;
    f32x2 adj[3][VP];
    {
        const int64_t am = p.adj_per_frame ? (int64_t)seg * p.frames + ta + jf : (int64_t)seg;
        const float *ab = p.ell_val + am * p.adj_seg_stride + jw * V;        // values of column w: (r V + w) V + v
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int i = 0; i < VP; ++i) {
                f32x2 a;
                if constexpr (EVEN) {
                    a = *reinterpret_cast<const f32x2 *>(ab + r * V * V + 2 * i);
                } else {                                   // odd V: columns start at odd offsets; the last pair has one joint
                    a[0] = ab[r * V * V + 2 * i];
                    a[1] = 2 * i + 1 < V ? ab[r * V * V + min(2 * i + 1, V - 1)] : 0.f;
                }
                adj[r][i] = jv ? a : f32x2{0.f, 0.f};
            }
    }

    f32x16 acc[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int g = 0; g < 16; ++g) acc[i][g] = 0.f;

    const float *seg_base = p.x + (int64_t)seg * p.x_seg_stride;
    const float *wbase = p.w + m0;
    // staging registers + chunk-invariant offsets (out-of-range slots clamped: they re-stage the last element)
    f32x4 wv[WB];
    unsigned wgo[WB], wlo[WB];
#pragma unroll
    for (int u = 0; u < WB; ++u) {
        const int e = min(u * NTHREADS + tid, R * KC2 * M4 - 1);
        const int row = e / M4, m4 = e % M4;
        wgo[u] = (unsigned)(((row / KC2) * p.CinPad + (row % KC2)) * p.Mpad + m4 * 4);
        wlo[u] = (unsigned)(e * 4);
    }
    // activation staging: joint pairs (8-byte loads; even V: every frame of a row starts 8-byte aligned) or single joints
    typedef std::conditional_t<EVEN, f32x2, float> xs_t;
    xs_t xv2[XB];
    unsigned xgo, xlo;
    constexpr int RPS = NTHREADS / XSL;                  // rows per sweep
    const int xrow0 = tid / XSL;                         // row of the first sweep; sweep u stages row xrow0 + RPS u
    if constexpr (EVEN) {
        const int pr = min(tid % XSL, fcnt * VP - 1);    // pair slot -> (frame, joint pair)
        const int f = pr / VP, w2 = pr - f * VP;
        xgo = (unsigned)(q0 + 2 * pr);
        xlo = (unsigned)(f * VPAD + 2 * w2);
    } else {
        const int e = min(tid % XSL, ncol - 1);          // joint slot -> (frame, joint)
        const int f = e / V;
        xgo = (unsigned)(q0 + e);
        xlo = (unsigned)(f * VPAD + (e - f * V));
    }
    auto issue_one = [&](int i, int c0) {
        if (i < WB) {
            wv[i] = *reinterpret_cast<const f32x4 *>(wbase + (size_t)c0 * p.Mpad + wgo[i]);
        } else {
            const int c = min(c0 + xrow0 + RPS * (i - WB), p.Cin - 1);       // clamped: padding channels carry zero weights
            xv2[i - WB] = *reinterpret_cast<const xs_t *>(seg_base + (int64_t)c * p.x_chan_stride + xgo);
        }
    };
    auto commit = [&](float *buf) {
#pragma unroll
        for (int u = 0; u < WB; ++u) *reinterpret_cast<f32x4 *>(buf + wlo[u]) = wv[u];
#pragma unroll
        for (int u = 0; u < XB; ++u) *reinterpret_cast<xs_t *>(buf + WSZ + (xrow0 + RPS * u) * LDX + xlo) = xv2[u];
    };
    if constexpr (!EVEN) {
        // odd V: the half-empty last joint pair multiplies the frame's first padding slot by a zero weight -- keep it finite
        for (int e = tid; e < 2 * KC2 * FT; e += NTHREADS) {
            const int b = e / (KC2 * FT), r = e % (KC2 * FT);
            smem[b * BUFSZ + WSZ + (r / FT) * LDX + (r % FT) * VPAD + V] = 0.f;
        }
    }
    // One MFMA k-step (operands in registers) with the NEXT k-step's operands formed between its MFMAs: a wave issues in
    // order, so the vector-ALU / LDS work has to sit between the MFMAs in program order to run in their shadow
    // (sched_barrier pins the order; left to itself the scheduler emits the MFMAs back to back and the rest behind them).
    // Slice 0: the lane's x row (VPAD / 4 reads); slice 1: the A operands; slices 2 .. M-2: the packed FMAs; slice M-1: the
    // final adds.  `nb` / `ns`: buffer and k-step the next operands come from.
    constexpr int M = R * NB, NSL = M - 3;
    auto step = [&](const float *nb, int ns, const float (&a_c)[R][NB], const float (&b_c)[R], float (&a_n)[R][NB], float (&b_n)[R]) {
        const float *bx = nb + WSZ + (2 * ns + kh) * LDX + xoff;
        const float *wr = nb + (2 * ns + kh) * MT + NB * l31;
        f32x4 xq[VPAD / 4];
        f32x2 sum[3] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
        for (int m = 0; m < M; ++m) {
            acc[m % NB] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_c[m / NB][m % NB], b_c[m / NB], acc[m % NB], 0, 0, 0);
            if (m == 0) {
#pragma unroll
                for (int i = 0; i < VPAD / 4; ++i) xq[i] = *reinterpret_cast<const f32x4 *>(bx + 4 * i);
                if constexpr (CONVRES) b_n[3] = bx[jw];
            }
            if (m == 1) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if constexpr (NB == 4) {
                        const f32x4 a = *reinterpret_cast<const f32x4 *>(wr + r * KC2 * MT);
#pragma unroll
                        for (int i = 0; i < 4; ++i) a_n[r][i] = a[i];
                    } else {
                        const f32x2 a = *reinterpret_cast<const f32x2 *>(wr + r * KC2 * MT);
#pragma unroll
                        for (int i = 0; i < 2; ++i) a_n[r][i] = a[i];
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < VP; ++i) {
                if (2 + i * NSL / VP == m) {
                    const f32x2 xp = {xq[i / 2][2 * (i & 1)], xq[i / 2][2 * (i & 1) + 1]};
#pragma unroll
                    for (int r = 0; r < 3; ++r) sum[r] = __builtin_elementwise_fma(xp, adj[r][i], sum[r]);
                }
            }
            if (m == M - 1) {
                // (the empty asm makes the sums opaque HERE: otherwise the SLP vectoriser pairs these adds with the ones of
                // the following k-step and this k-step's MFMAs end up depending on the next one's FMA chains)
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    b_n[r] = sum[r][0] + sum[r][1];
                    asm volatile("" : "+v"(b_n[r]));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // the same operands without MFMAs beside them (first k-step of the kernel)
    auto form = [&](const float *nb, int ns, float (&a_n)[R][NB], float (&b_n)[R]) {
        const float *bx = nb + WSZ + (2 * ns + kh) * LDX + xoff;
        const float *wr = nb + (2 * ns + kh) * MT + NB * l31;
        f32x4 xq[VPAD / 4];
#pragma unroll
        for (int i = 0; i < VPAD / 4; ++i) xq[i] = *reinterpret_cast<const f32x4 *>(bx + 4 * i);
        if constexpr (CONVRES) b_n[3] = bx[jw];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int i = 0; i < NB; ++i) a_n[r][i] = wr[r * KC2 * MT + i];
        f32x2 sum[3] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
        for (int i = 0; i < VP; ++i) {
            const f32x2 xp = {xq[i / 2][2 * (i & 1)], xq[i / 2][2 * (i & 1) + 1]};
#pragma unroll
            for (int r = 0; r < 3; ++r) sum[r] = __builtin_elementwise_fma(xp, adj[r][i], sum[r]);
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            b_n[r] = sum[r][0] + sum[r][1];
            asm volatile("" : "+v"(b_n[r]));
        }
    };

    // K loop.  One barrier per chunk, placed in FRONT of the chunk's last k-step: by then that k-step's operands are in
    // registers (nothing reads `cur` any more) and chunk c+1 is committed in `oth`, so the last k-step's MFMAs run beside
    // the forming of the next chunk's first operands -- the pipeline runs through the chunk boundary.
    const int nchunks = p.CinPad / KC2;
#pragma unroll
    for (int i = 0; i < NL; ++i) issue_one(i, 0);
    commit(smem);
    if (nchunks > 1) {
#pragma unroll
        for (int i = 0; i < NL; ++i) issue_one(i, KC2);
    }
    __syncthreads();
    float aq[2][R][NB], bq[2][R];
    form(smem, 0, aq[0], bq[0]);
    for (int c = 0; c < nchunks; ++c) {
        float *cur = smem + (c & 1) * BUFSZ, *oth = smem + ((c & 1) ^ 1) * BUFSZ;
        const bool more = c + 1 < nchunks;
        if (more) commit(oth);                             // chunk c+1: registers -> the other buffer
        const int cnext = min(c + 2, nchunks - 1) * KC2;   // (past the end: the last chunk again, into dead registers)
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (s < NH) {
#pragma unroll
                for (int i = s * NL / NH; i < (s + 1) * NL / NH; ++i) issue_one(i, cnext);
            }
            if (s == NS - 1) {
                __builtin_amdgcn_s_setprio(0);
                __syncthreads();                           // (unconditional: a branch here lets the compiler sink the
                                                           //  forming of the last k-step's operands below the barrier)
                __builtin_amdgcn_s_setprio(1);
            }
            // (last k-step of the last chunk: the "next" operands are formed from `cur` again and never used)
            step(s == NS - 1 && more ? oth : cur, (s + 1) % NS, aq[s & 1], bq[s & 1], aq[(s + 1) & 1], bq[(s + 1) & 1]);
        }
        __builtin_amdgcn_s_setprio(0);
    }

    // epilogue: ReLU(acc + bias + identity residual).  acc[i][g] of lane (column j, half kh) is output row
    // m0 + NB ((g & 3) + 8 (g >> 2) + 4 kh) + i; a store instruction writes two 128-byte row segments.
    float *oseg = p.y + (int64_t)seg * p.y_seg_stride;
    const bool ident = p.res_mode == CSK_RES_IDENTITY;
    const bool full = p.fast_epi && m0 + MT <= p.Cout;
    const int qc = min(q0 + j, Q - 1);
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        float bb[16], rv[16];
        if (full) {
            const unsigned lb = 4u * (unsigned)(NB * 4 * kh);
            const unsigned lx = 4u * ((unsigned)(NB * 4 * kh) * (unsigned)p.x_chan_stride + (unsigned)qc);
            const unsigned ly = 4u * ((unsigned)(NB * 4 * kh) * (unsigned)p.y_chan_stride + (unsigned)(q0 + j));
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int row = m0 + NB * ((g & 3) + 8 * (g >> 2)) + i;
                bb[g] = ld_lane(p.bias + row, lb);
                rv[g] = ident ? ld_lane(seg_base + (int64_t)row * p.x_chan_stride, lx) : 0.f;
            }
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[i][g] = relu_nan(acc[i][g] + bb[g] + rv[g]);
            if (jv) {
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const int row = m0 + NB * ((g & 3) + 8 * (g >> 2)) + i;
                    st_lane(oseg + (int64_t)row * p.y_chan_stride, ly, acc[i][g]);
                }
            }
        } else {
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int row = m0 + NB * ((g & 3) + 8 * (g >> 2) + 4 * kh) + i;
                const float bv = p.bias[row];              // bias is padded to Mpad
                const float r0 = ident ? seg_base[(int64_t)min(row, p.Cout - 1) * p.x_chan_stride + qc] : 0.f;
                const float v = relu_nan(acc[i][g] + bv + r0);
                if (jv && row < p.Cout) oseg[(int64_t)row * p.y_chan_stride + q0 + j] = v;
            }
        }
    }
}

template <int MT, bool CONVRES, int V, int KC2>
static int launch_dense2(const GcnParams &p, int n_seg, hipStream_t stream) {
    constexpr int VPAD = (V + 3) & ~3, FT = 128 / V, R = CONVRES ? 4 : 3;
    const size_t lds = 2 * (size_t)(R * KC2 * MT + KC2 * FT * VPAD) * sizeof(float);
    void (*k)(GcnParams) = gcn_stage_dense2_kernel<MT, CONVRES, V, KC2>;
    if (const int e = csk_ensure_lds((const void *)k, lds)) return e;
    hipLaunchKernelGGL(k, dim3(p.qtiles * p.mtiles * n_seg), dim3(NTHREADS), lds, stream, p);
    return (int)hipGetLastError();
}

template <int V>
static int dispatch_dense2(GcnParams p, int n_seg, hipStream_t s) {
    constexpr int FT = 128 / V;
    // (64-row tiles -- three workgroups per CU -- for the under-filled step launches of the 128- / 256-row layers: 1 014-1 016 k
    // against 1 028 k frames/s, one shard 896-898 k against 891 k: noise, not adopted)
    const bool big = (p.Mpad % 128) == 0;
    const int MT = big ? 128 : 64;
    p.lds_frames = FT;
    p.qtiles = (p.frames + FT - 1) / FT;
    p.mtiles = p.Mpad / MT;
    if ((int64_t)p.qtiles * p.mtiles * n_seg >= (1ll << 31)) return -2;
    // (16-channel chunks -- one barrier per 96 / 48 MFMAs -- measured equal on clips and 3 % slower online: 8 it is)
    if (big) return p.R == 4 ? launch_dense2<128, true, V, 8>(p, n_seg, s) : launch_dense2<128, false, V, 8>(p, n_seg, s);
    return p.R == 4 ? launch_dense2<64, true, V, 8>(p, n_seg, s) : launch_dense2<64, false, V, 8>(p, n_seg, s);
}

int csk_launch_gcn_dense2(GcnParams p, int n_seg, void *stream) {
    // built for the two skeleton layouts of the reference's datasets: V = 18 (Kinetics / OpenPose) and V = 25 (NTU RGB+D)
    if (!p.dense || (p.V != 18 && p.V != 25) || p.frames < 1) return -2;
    if (p.V % 2 == 0) {   // joint pairs are staged with 8-byte loads
        if ((reinterpret_cast<uintptr_t>(p.x) & 7) || (p.x_seg_stride & 1) || (p.x_chan_stride & 1)) return -2;
        if ((reinterpret_cast<uintptr_t>(p.ell_val) & 7) || (p.adj_seg_stride & 1)) return -2;
    }
    hipStream_t s = (hipStream_t)stream;
    return p.V == 18 ? dispatch_dense2<18>(p, n_seg, s) : dispatch_dense2<25>(p, n_seg, s);
}
