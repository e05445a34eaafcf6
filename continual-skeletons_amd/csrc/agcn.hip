// agcn.hip -- A-GCN adaptive adjacency (models/a_gcn/a_gcn.py:53-63) on gfx950.
#include <type_traits>

#include "mfma_core.h"

// Softmax arithmetic shared by every attention kernel of this file (so that the fused and the two-launch routes, and the
// per-frame and per-segment forms, keep producing the same bits): the exponential through the hardware v_exp_f32 (a multiply
// by log2 e in front; arguments are <= 0, relative error of the result <= |x| * 2^-23 -- 1e-6 at x = -10) and ONE IEEE
// division per column (reciprocal of the column sum) instead of one per element.  With libm's expf and per-element IEEE
// divisions the softmax was ~3 k of the ~4 k cycles a wave spent on one (pair, skeleton) attention unit of the fused
// step kernel (16 exponentials + 16 divisions of ~10-15 instructions each per lane): profiles/HISTORY.md round 4.
__device__ __forceinline__ float sm_exp(float x) { return __expf(x); }
__device__ __forceinline__ float sm_rcp(float sum) { return 1.f / sum; }

// ------------------------------------------------------------------------------------------------
// A-GCN attention (models/a_gcn/a_gcn.py:53-63): per sample n and subset i
//     logits[v, w] = sum_{k,t} Ea[i][k][t][v] * Eb[i][k][t][w] / (inter * T)
//     adj[i][v, w] = softmax over v (dim -2) of logits + (A + graph_attn)[i][v, w]
// E = (n_seg, 6*inter, T, V): channels [i*inter + k] = a_conv_i, [3*inter + i*inter + k] = b_conv_i (biases
// included), produced by csk_tcn_stage_f32 as a 1x1 conv.  Output: the column-wise dense ELL values
// ell_val[n][i][w][v] = adj[i][v, w] consumed by gcn_stage_kernel with adj_seg_stride = 3*V*V.
// ------------------------------------------------------------------------------------------------
// Clip form (T > 1), two launches:
//  (1) agcn_logits_partial_kernel, grid (n, subset, KSPL channel ranges): logits = Ea^T . Eb over K = inter * T rows as an
//      fp32-MFMA product, A[i = v][k] = Ea[row k][v], B[k][j = w] = Eb[row k][w] (V <= 32 columns used).  A channel of E is
//      T * V contiguous floats, so rows are streamed in chunks of 64 frames (64 * V contiguous floats per operand) with
//      16-byte loads, register-prefetched one chunk ahead, into LDS; the four waves take 16 rows of a chunk each and
//      their partial 32 x 32 tiles are summed through LDS in a fixed order.  (The first form -- one workgroup per
//      (n, subset), operands read from global memory 4 bytes per lane with 18 of 32 lanes active -- ran at 2 TB/s:
//      1.6 ms of a 20.6 ms A-GCN forward, profiles/r03a_agcn_clip_layers.md.)
//  (2) agcn_softmax_kernel: sums the KSPL partials in a fixed order, softmax over v, + (A + graph_attn).
// KSPL is a constant of the kernel (not of the batch size): a sample's result does not depend on its batch.
static constexpr int ATT_KSPL = 4;      // channel ranges per (sample, subset)
static constexpr int ATT_FR = 64;       // frames per staged chunk

__global__ __launch_bounds__(256) void agcn_logits_partial_kernel(const float *__restrict__ E, float *__restrict__ part,
                                                                  int inter, int T, int V, int64_t e_seg_stride,
                                                                  int64_t e_chan_stride, int seg_per_group, int64_t e_group_stride) {
    extern __shared__ __attribute__((aligned(16))) float lds_att[];       // [2 operands][ATT_FR * V + 64] then [4][32][33]
    const int n = blockIdx.x, i = blockIdx.y, ks = blockIdx.z, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5;
    const int opsz = (ATT_FR * V + 64 + 3) & ~3;                           // floats per operand buffer (+ slack for lanes >= V), 16-byte multiple
    float *La = lds_att, *Lb = lds_att + opsz, *red = lds_att + 2 * opsz;
    const int cper = (inter + ATT_KSPL - 1) / ATT_KSPL;
    const int k0 = ks * cper, k1 = min(inter, k0 + cper);
    const float *eseg = E + (int64_t)(n / seg_per_group) * e_group_stride + (int64_t)(n % seg_per_group) * e_seg_stride;
    const float *ea = eseg + (int64_t)i * inter * e_chan_stride;
    const float *eb = eseg + (int64_t)(3 + i) * inter * e_chan_stride;
    f32x16 acc;
#pragma unroll
    for (int g = 0; g < 16; ++g) acc[g] = 0.f;
    const int nfr = (T + ATT_FR - 1) / ATT_FR;                             // chunks per channel
    const int nchunk = max(k1 - k0, 0) * nfr;
    constexpr int NV4 = 2;                                                 // float4 per thread and operand: 2 * 256 * 4 >= 64 * 32
    f32x4 va[NV4], vb[NV4];
    auto issue = [&](int c) {
        const int kk = k0 + c / nfr, t0 = (c % nfr) * ATT_FR;
        const int nflt = min(ATT_FR, T - t0) * V;                          // floats of this chunk
        const float *pa = ea + (int64_t)kk * e_chan_stride + (int64_t)t0 * V, *pb = eb + (int64_t)kk * e_chan_stride + (int64_t)t0 * V;
#pragma unroll
        for (int u = 0; u < NV4; ++u) {
            // clamped to the chunk's last (possibly partial) vector: duplicates land on themselves; the partial vector
            // reads up to 3 floats past the chunk -- rows that the MFMA loop masks (E is readable 12 bytes past its end)
            const int f = min(4 * (u * 256 + tid), (nflt - 1) & ~3);
            va[u] = *reinterpret_cast<const f32x4u *>(pa + f);
            vb[u] = *reinterpret_cast<const f32x4u *>(pb + f);
        }
    };
    auto commit = [&](int c) {
        const int t0 = (c % nfr) * ATT_FR;
        const int nflt = min(ATT_FR, T - t0) * V;
#pragma unroll
        for (int u = 0; u < NV4; ++u) {
            const int f = min(4 * (u * 256 + tid), (nflt - 1) & ~3);
            *reinterpret_cast<f32x4 *>(La + f) = va[u];
            *reinterpret_cast<f32x4 *>(Lb + f) = vb[u];
        }
    };
    if (nchunk > 0) issue(0);
    for (int c = 0; c < nchunk; ++c) {
        __syncthreads();                                                   // the previous chunk's reads are done
        commit(c);
        __syncthreads();
        if (c + 1 < nchunk) issue(c + 1);
        const int rows = min(ATT_FR, T - (c % nfr) * ATT_FR);
#pragma unroll
        for (int s = 0; s < ATT_FR / 8; ++s) {                             // 16 rows per wave = 8 k-steps of 2 rows
            const int r = wave * (ATT_FR / 4) + 2 * s + kh;
            const float a = La[r * V + l31], b = Lb[r * V + l31];
            const bool ok = r < rows;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ok ? a : 0.f, ok ? b : 0.f, acc, 0, 0, 0);
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh) * 33 + l31] = acc[r];   // [wave][v][w]
    __syncthreads();
    float *dst = part + ((int64_t)(n * 3 + i) * ATT_KSPL + ks) * V * V;
    for (int e = tid; e < V * V; e += 256) {
        const int v = e / V, w = e - v * V;
        dst[e] = ((red[(0 * 32 + v) * 33 + w] + red[(1 * 32 + v) * 33 + w]) + red[(2 * 32 + v) * 33 + w]) + red[(3 * 32 + v) * 33 + w];
    }
}

// adj[n][i][w][v] = softmax over v of (sum_ks part[n][i][ks][v][w]) / K + (A + graph_attn)[i][v][w]; one wave per (n, i)
__global__ __launch_bounds__(256) void agcn_softmax_kernel(const float *__restrict__ part, const float *__restrict__ a_sum,
                                                           float *__restrict__ ell_val, int n_pairs, int K, int V) {
    const int lane = threadIdx.x & 63;
    const int pair = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pair >= n_pairs || lane >= V) return;
    const int i = pair % 3, w = lane;
    const float *src = part + (int64_t)pair * ATT_KSPL * V * V;
    float lg[32];
    float m = -INFINITY;
    for (int v = 0; v < V; ++v) {
        float s = 0.f;
#pragma unroll
        for (int ks = 0; ks < ATT_KSPL; ++ks) s += src[(ks * V + v) * V + w];
        lg[v] = s / (float)K;
        m = fmaxf(m, lg[v]);
    }
    float sum = 0.f;
    for (int v = 0; v < V; ++v) sum += sm_exp(lg[v] - m);
    float *dst = ell_val + ((int64_t)pair * V + w) * V;
    const float rs = sm_rcp(sum);
    for (int v = 0; v < V; ++v) dst[v] = sm_exp(lg[v] - m) * rs + a_sum[(i * V + v) * V + w];
}

// ------------------------------------------------------------------------------------------------
// T == 1 (CoAGCN, coa_gcn.py: the module is applied per frame): K = inter rows only, so one WAVE handles one
// (skeleton, subset) pair end to end -- K/2 MFMAs, the softmax over v in registers (a column w lives in lanes w and
// w + 32: 16 rows each), no LDS, no barrier.  256-thread workgroups = 4 pairs.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void agcn_attention_step_kernel(const float *__restrict__ E, const float *__restrict__ a_sum,
                                                                  float *__restrict__ ell_val, int n_pairs, int inter, int V,
                                                                  int64_t e_seg_stride, int64_t e_chan_stride,
                                                                  int seg_per_group, int64_t e_group_stride) {
    const int lane = threadIdx.x & 63, l31 = lane & 31, kh = lane >> 5;
    const int pair = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (pair >= n_pairs) return;
    const int n = pair / 3, i = pair - 3 * n;
    const float *eseg = E + (int64_t)(n / seg_per_group) * e_group_stride + (int64_t)(n % seg_per_group) * e_seg_stride;
    const float *ea = eseg + (int64_t)i * inter * e_chan_stride + min(l31, V - 1);
    const float *eb = eseg + (int64_t)(3 + i) * inter * e_chan_stride + min(l31, V - 1);
    const bool col = l31 < V;
    f32x16 acc;
#pragma unroll
    for (int g = 0; g < 16; ++g) acc[g] = 0.f;
    constexpr int UN = 8;                                   // 16 rows per batch of loads
    for (int k0 = 0; k0 < inter; k0 += 2 * UN) {
        float av[UN], bv[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int k = k0 + 2 * u + kh;
            const int64_t off = (int64_t)min(k, inter - 1) * e_chan_stride;
            const float xa = ea[off], xb = eb[off];
            av[u] = (col && k < inter) ? xa : 0.f;
            bv[u] = (col && k < inter) ? xb : 0.f;
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
    }
    // acc[r] = logits[v][w] * inter with v = (r & 3) + 8 (r >> 2) + 4 kh, w = l31; softmax over v < V
    const float inv = 1.f / (float)inter;
    float m = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int v = (r & 3) + 8 * (r >> 2) + 4 * kh;
        acc[r] *= inv;
        if (v < V) m = fmaxf(m, acc[r]);
    }
    m = fmaxf(m, __shfl_xor(m, 32));
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int v = (r & 3) + 8 * (r >> 2) + 4 * kh;
        acc[r] = v < V ? sm_exp(acc[r] - m) : 0.f;
        sum += acc[r];
    }
    sum += __shfl_xor(sum, 32);
    const float rs = sm_rcp(sum);
    if (col) {
        float *dst = ell_val + ((int64_t)(n * 3 + i) * V + l31) * V;            // [n][i][w][v]
        const float *as = a_sum + (int64_t)i * V * V + l31;                       // [i][v][w]
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int v = (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (v < V) dst[v] = acc[r] * rs + as[v * V];
        }
    }
}

extern "C" int csk_agcn_attention_f32(const float *E, const float *a_sum, float *ell_val, float *scratch, int n_seg, int inter,
                                      int T, int V, int64_t e_seg_stride, int64_t e_chan_stride, int seg_per_group,
                                      int64_t e_group_stride, void *stream) {
    if (!E || !a_sum || !ell_val) CSK_FAIL("agcn_attention: null pointer");
    if (n_seg <= 0 || inter <= 0 || T <= 0 || V < 2 || V > 32) CSK_FAIL("agcn_attention: bad dims (V <= 32)");
    if (seg_per_group <= 0) CSK_FAIL("agcn_attention: seg_per_group must be positive");
    if (T == 1) {
        const int pairs = 3 * n_seg;
        hipLaunchKernelGGL(agcn_attention_step_kernel, dim3((pairs + 3) / 4), dim3(256), 0, (hipStream_t)stream, E, a_sum,
                           ell_val, pairs, inter, V, e_seg_stride, e_chan_stride, seg_per_group, e_group_stride);
        return (int)hipGetLastError();
    }
    if (!scratch) CSK_FAIL("agcn_attention: the clip form needs a scratch buffer of n_seg * 3 * %d * V * V floats", ATT_KSPL);
    const size_t lds = (size_t)(2 * ((ATT_FR * V + 64 + 3) & ~3) + 4 * 32 * 33) * sizeof(float);
    if (const int e = csk_ensure_lds((const void *)agcn_logits_partial_kernel, lds)) return e;
    hipLaunchKernelGGL(agcn_logits_partial_kernel, dim3(n_seg, 3, ATT_KSPL), dim3(256), lds, (hipStream_t)stream, E, scratch, inter, T,
                       V, e_seg_stride, e_chan_stride, seg_per_group, e_group_stride);
    if (const int e = (int)hipGetLastError()) return e;
    hipLaunchKernelGGL(agcn_softmax_kernel, dim3((3 * n_seg + 3) / 4), dim3(256), 0, (hipStream_t)stream, scratch, a_sum, ell_val,
                       3 * n_seg, inter * T, V);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Fused form: embedding 1x1 convs + attention logits in ONE launch (no E tensor in memory); step form (CoAGCN, one
// attention per skeleton frame) and clip form (A-GCN, one attention per segment: partial logits per tile + a softmax launch).
// A workgroup takes FT = 128 / V whole skeletons ("frames" of a channel-major slot) and NP (a_i, b_i) pairs:
//   (1) E[rows][cols] = W_e . x + b_e for its 2 * INTER * NP embedding rows (pair-major row order: a_i rows, then b_i rows)
//       as an fp32-MFMA GEMM over C_in -- a wave owns all rows x 32 columns, accumulator block i / MFMA row rho = tile row
//       NB rho + i (one vector LDS read per A operand), ping-pong LDS, register prefetch, one barrier per 16-channel chunk;
//   (2) the E tile goes to LDS; per (pair, skeleton) one wave forms logits = Ea^T . Eb (INTER / 2 MFMAs), the softmax over v
//       in registers and writes adj[skeleton][i][w][v] -- the arithmetic of agcn_attention_step_kernel above, operand for
//       operand (the E values equal csk_conv1x1_f32's: channels are accumulated in the same order).
// V in {18, 25} (even V: pairs of joints are the staging unit; odd V: single joints), INTER in {16, 32, 64}; other shapes take
// the two-launch route.
// ------------------------------------------------------------------------------------------------
struct EmbAttParams {
    const float *x, *w, *bias, *a_sum;
    float *ell_val, *part;
    int64_t x_seg_stride, x_chan_stride;
    int Cin, CinPad, Mpad, frames;
    unsigned qtiles, mtiles;
};
typedef float f32x2a __attribute__((ext_vector_type(2)));

template <int INTER, int NP, int V, bool CLIP>
__global__ __launch_bounds__(NTHREADS, 2) void agcn_embed_attention_kernel(const EmbAttParams p) {
    constexpr int VP = V / 2, VPAD = (V + 3) & ~3, NT = 128, FT = NT / V;
    constexpr bool EVEN = V % 2 == 0;                     // joint pairs staged with 8-byte loads; odd V: joint by joint
    constexpr int XSL = EVEN ? 64 : 128, RPS = NTHREADS / XSL;
    constexpr int MT = 2 * INTER * NP, NB = MT / 32;
    constexpr int KC2 = 16, NS = KC2 / 2, NH = NS / 2;
    constexpr int LDX = FT * VPAD, M4 = MT / 4;
    constexpr int WB = (KC2 * M4 + NTHREADS - 1) / NTHREADS, XB = KC2 * XSL / NTHREADS, NL = WB + XB;
    constexpr int WSZ = KC2 * MT, BUFSZ = WSZ + KC2 * LDX;
    constexpr int ESLD = NT + 4;                          // E tile row stride
    static_assert(MT % 32 == 0 && FT * V <= NT && V <= 32 && INTER % 2 == 0, "tile shape");
    extern __shared__ __attribute__((aligned(16))) float smem_ea[];      // max(2 * BUFSZ, MT * ESLD) floats

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5;
    const unsigned wid = xcd_contiguous_id(blockIdx.x, gridDim.x);
    const int mt = (int)(wid % p.mtiles), qt = (int)((wid / p.mtiles) % p.qtiles);
    const int seg = (int)(wid / (p.mtiles * p.qtiles));
    const int m0 = mt * MT;
    const int ta = qt * FT, q0 = ta * V;
    const int fcnt = min(FT, p.frames - ta);
    const int ncol = fcnt * V;
    const int j = wave * 32 + l31;
    const bool jv = j < ncol;
    const int jf = jv ? j / V : 0, jw = jv ? j - jf * V : 0;
    const int xcol = jf * VPAD + jw;

    f32x16 acc[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int g = 0; g < 16; ++g) acc[i][g] = 0.f;

    const float *seg_base = p.x + (int64_t)seg * p.x_seg_stride;
    const float *wbase = p.w + m0;
    f32x4 wv[WB];
    unsigned wgo[WB], wlo[WB];
#pragma unroll
    for (int u = 0; u < WB; ++u) {
        const int e = min(u * NTHREADS + tid, KC2 * M4 - 1);
        const int row = e / M4, m4 = e % M4;
        wgo[u] = (unsigned)(row * p.Mpad + m4 * 4);
        wlo[u] = (unsigned)(e * 4);
    }
    typedef std::conditional_t<EVEN, f32x2a, float> xs_t;
    xs_t xv2[XB];
    unsigned xgo, xlo;
    const int xrow0 = tid / XSL;
    if constexpr (EVEN) {
        const int pr = min(tid % XSL, fcnt * VP - 1);
        const int f = pr / VP, w2 = pr - f * VP;
        xgo = (unsigned)(q0 + 2 * pr);
        xlo = (unsigned)(f * VPAD + 2 * w2);
    } else {
        const int e = min(tid % XSL, ncol - 1);
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
    const int nchunks = p.CinPad / KC2;
#pragma unroll
    for (int i = 0; i < NL; ++i) issue_one(i, 0);
    commit(smem_ea);
    if (nchunks > 1) {
#pragma unroll
        for (int i = 0; i < NL; ++i) issue_one(i, KC2);
    }
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        float *cur = smem_ea + (c & 1) * BUFSZ, *oth = smem_ea + ((c & 1) ^ 1) * BUFSZ;
        if (c + 1 < nchunks) commit(oth);
        const int cnext = min(c + 2, nchunks - 1) * KC2;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (s < NH) {
#pragma unroll
                for (int i = s * NL / NH; i < (s + 1) * NL / NH; ++i) issue_one(i, cnext);
                // pinned to this k-step (left alone, the scheduler sinks the loads to the end of the block, right in front of
                // the barrier and the commit that waits for them)
                __builtin_amdgcn_sched_barrier(0);
            }
            const int kk = 2 * s + kh;
            const float b = cur[WSZ + kk * LDX + xcol];
            const float *wr = cur + kk * MT + NB * l31;
#pragma unroll
            for (int i = 0; i < NB; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[i], b, acc[i], 0, 0, 0);
        }
        __syncthreads();
    }
    // E tile (+ bias) -> LDS, row-major [tile row][column]
    float *Es = smem_ea;
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int row = NB * ((g & 3) + 8 * (g >> 2) + 4 * kh) + i;
            Es[row * ESLD + j] = acc[i][g] + p.bias[m0 + row];
        }
    __syncthreads();
    const int cl = min(l31, V - 1);
    if constexpr (CLIP) {
        // clip form: ONE attention per segment over all its frames -- this tile contributes the partial logits of its
        // frames, part[seg][pair][tile][v][w] = sum_{f, k} Ea[k][f, v] Eb[k][f, w]; agcn_softmax_parts_kernel sums the tiles
        // in order.  Wave q takes frames f = q, q + 4, ...; the four partial 32 x 32 tiles are added through LDS in a
        // fixed order (for NP = 1 the buffer aliases the E tile once everybody has read it).
        float *red = NP == 1 ? smem_ea : smem_ea + MT * ESLD;                  // [4][32][33]
#pragma unroll 1
        for (int pr = 0; pr < NP; ++pr) {
            f32x16 lg;
#pragma unroll
            for (int g = 0; g < 16; ++g) lg[g] = 0.f;
            for (int f = wave; f < fcnt; f += NTHREADS / 64) {
                const float *ea = Es + (pr * 2 * INTER + kh) * ESLD + f * V + cl;
                const float *eb = ea + INTER * ESLD;
#pragma unroll 8
                for (int s = 0; s < INTER / 2; ++s) lg = __builtin_amdgcn_mfma_f32_32x32x2f32(ea[2 * s * ESLD], eb[2 * s * ESLD], lg, 0, 0, 0);
            }
            if (NP == 1) __syncthreads();                                       // all reads of the E tile are done
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh) * 33 + l31] = lg[r];
            __syncthreads();
            const int pi = mt * NP + pr;
            float *dst = p.part + (((int64_t)seg * 3 + pi) * p.qtiles + qt) * (V * V);
            for (int e = tid; e < V * V; e += NTHREADS) {
                const int v = e / V, w = e - v * V;
                dst[e] = ((red[(0 * 32 + v) * 33 + w] + red[(1 * 32 + v) * 33 + w]) + red[(2 * 32 + v) * 33 + w]) + red[(3 * 32 + v) * 33 + w];
            }
            if (NP > 1) __syncthreads();                                        // red is rewritten by the next pair
        }
    } else {
    // step form: attention units (pair, skeleton) dealt to the waves
    const float inv = 1.f / (float)INTER;
    const bool col = l31 < V;
    for (int u = wave; u < NP * fcnt; u += NTHREADS / 64) {
        const int pr = u / fcnt, f = u - pr * fcnt;
        const float *ea = Es + (pr * 2 * INTER + kh) * ESLD + f * V + cl;
        const float *eb = ea + INTER * ESLD;
        f32x16 lg;
#pragma unroll
        for (int g = 0; g < 16; ++g) lg[g] = 0.f;
#pragma unroll 8
        for (int s = 0; s < INTER / 2; ++s) lg = __builtin_amdgcn_mfma_f32_32x32x2f32(ea[2 * s * ESLD], eb[2 * s * ESLD], lg, 0, 0, 0);
        float m = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int v = (r & 3) + 8 * (r >> 2) + 4 * kh;
            lg[r] *= inv;
            if (v < V) m = fmaxf(m, lg[r]);
        }
        m = fmaxf(m, __shfl_xor(m, 32));
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int v = (r & 3) + 8 * (r >> 2) + 4 * kh;
            lg[r] = v < V ? sm_exp(lg[r] - m) : 0.f;
            sum += lg[r];
        }
        sum += __shfl_xor(sum, 32);
        const float rs = sm_rcp(sum);
        if (col) {
            const int pi = mt * NP + pr;
            const int64_t n = (int64_t)seg * p.frames + ta + f;
            float *dst = p.ell_val + ((n * 3 + pi) * V + l31) * V;               // [n][i][w][v]
            const float *as = p.a_sum + (int64_t)pi * V * V + l31;               // [i][v][w]
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int v = (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (v < V) dst[v] = lg[r] * rs + as[v * V];
            }
        }
    }
    }
}

// adj[n][i][w][v] = softmax over v of (sum_tiles part[n][i][tile][v][w]) / K + (A + graph_attn)[i][v][w]; one workgroup per
// (n, i): thread e = (v, w) sums its element over the tiles in order (coalesced), then lanes w < V do the softmax from LDS
__global__ __launch_bounds__(NTHREADS) void agcn_softmax_parts_kernel(const float *__restrict__ part, const float *__restrict__ a_sum,
                                                                      float *__restrict__ ell_val, int nparts, int K, int V) {
    __shared__ float lg[32 * 33];
    const int pair = blockIdx.x, i = pair % 3, tid = threadIdx.x;
    const float *src = part + (int64_t)pair * nparts * V * V;
    for (int e = tid; e < V * V; e += NTHREADS) {
        float s = 0.f;
        // eight tiles' partial sums in flight per thread (a rolled loop waits for every load: ~43 dependent round trips at T = 300);
        // the additions stay in tile order
        for (int t0 = 0; t0 < nparts; t0 += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)min(t0 + u, nparts - 1) * V * V + e];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (t0 + u < nparts) s += v[u];
        }
        lg[(e / V) * 33 + e % V] = s / (float)K;
    }
    __syncthreads();
    if (tid < V) {
        const int w = tid;
        float m = -INFINITY;
        for (int v = 0; v < V; ++v) m = fmaxf(m, lg[v * 33 + w]);
        float sum = 0.f;
        for (int v = 0; v < V; ++v) sum += sm_exp(lg[v * 33 + w] - m);
        const float rs = sm_rcp(sum);
        float *dst = ell_val + ((int64_t)pair * V + w) * V;
        for (int v = 0; v < V; ++v) dst[v] = sm_exp(lg[v * 33 + w] - m) * rs + a_sum[(i * V + v) * V + w];
    }
}

template <int INTER, int NP, int V, bool CLIP>
static int launch_embed_attention(EmbAttParams p, int n_seg, hipStream_t stream) {
    constexpr int VPAD = (V + 3) & ~3, FT = 128 / V, MT = 2 * INTER * NP;
    p.mtiles = 3 / NP;
    p.qtiles = (p.frames + FT - 1) / FT;
    const size_t a = 2 * (size_t)(16 * MT + 16 * FT * VPAD), b = (size_t)MT * (128 + 4) + (CLIP && NP > 1 ? 4 * 32 * 33 : 0);
    const size_t lds = (a > b ? a : b) * sizeof(float);
    void (*k)(EmbAttParams) = agcn_embed_attention_kernel<INTER, NP, V, CLIP>;
    if (const int e = csk_ensure_lds((const void *)k, lds)) return e;
    hipLaunchKernelGGL(k, dim3(p.qtiles * p.mtiles * n_seg), dim3(NTHREADS), lds, stream, p);
    return (int)hipGetLastError();
}

extern "C" int csk_agcn_embed_attention_f32(const float *x, const float *w_pairs, const float *b_pairs, const float *a_sum,
                                            float *ell_val, float *scratch, int n_seg, int c_in, int inter, int frames, int V,
                                            int per_frame, int64_t x_seg_stride, int64_t x_chan_stride, void *stream) {
    if (!x || !w_pairs || !b_pairs || !a_sum || !ell_val) CSK_FAIL("agcn_embed_attention: null pointer");
    if (n_seg <= 0 || c_in <= 0 || frames <= 0) CSK_FAIL("agcn_embed_attention: bad dims");
    if ((V != 18 && V != 25) || (inter != 16 && inter != 32 && inter != 64))
        CSK_FAIL("agcn_embed_attention: built for V in {18, 25} and inter in {16, 32, 64} (use csk_conv1x1_f32 + csk_agcn_attention_f32)");
    if (V % 2 == 0 && ((reinterpret_cast<uintptr_t>(x) & 7) || (x_seg_stride & 1) || (x_chan_stride & 1)))
        CSK_FAIL("agcn_embed_attention: activation rows must be 8-byte aligned");
    if ((int64_t)n_seg * frames >= (1ll << 31) / (3 * V * V)) CSK_FAIL("agcn_embed_attention: too many skeletons for one launch");
    if (!per_frame && !scratch)
        CSK_FAIL("agcn_embed_attention: the per-segment form needs n_seg * 3 * ceil(frames / (128 / V)) * V * V floats of scratch");
    EmbAttParams p;
    p.x = x; p.w = w_pairs; p.bias = b_pairs; p.a_sum = a_sum; p.ell_val = ell_val; p.part = scratch;
    p.x_seg_stride = x_seg_stride; p.x_chan_stride = x_chan_stride;
    p.Cin = c_in; p.CinPad = round_up(c_in, CSK_CPAD); p.Mpad = round_up(6 * inter, CSK_MT); p.frames = frames;
    hipStream_t s = (hipStream_t)stream;
    int rc;
#define CSK_EA_LAUNCH(VV, CL)                                                                            \
    (inter == 16 ? launch_embed_attention<16, 3, VV, CL>(p, n_seg, s)                                    \
                 : inter == 32 ? launch_embed_attention<32, 1, VV, CL>(p, n_seg, s) : launch_embed_attention<64, 1, VV, CL>(p, n_seg, s))
    if (per_frame) return V == 18 ? CSK_EA_LAUNCH(18, false) : CSK_EA_LAUNCH(25, false);
    rc = V == 18 ? CSK_EA_LAUNCH(18, true) : CSK_EA_LAUNCH(25, true);
#undef CSK_EA_LAUNCH
    if (rc) return rc;
    const int nparts = (frames + 128 / V - 1) / (128 / V);
    hipLaunchKernelGGL(agcn_softmax_parts_kernel, dim3(3 * n_seg), dim3(NTHREADS), 0, s, scratch, a_sum, ell_val, nparts,
                       inter * frames, V);
    return (int)hipGetLastError();
}
