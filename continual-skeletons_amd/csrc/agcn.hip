// agcn.hip -- A-GCN adaptive adjacency (models/a_gcn/a_gcn.py:53-63) on gfx950.
#include "mfma_core.h"

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
    for (int v = 0; v < V; ++v) sum += expf(lg[v] - m);
    float *dst = ell_val + ((int64_t)pair * V + w) * V;
    for (int v = 0; v < V; ++v) dst[v] = expf(lg[v] - m) / sum + a_sum[(i * V + v) * V + w];
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
        acc[r] = v < V ? expf(acc[r] - m) : 0.f;
        sum += acc[r];
    }
    sum += __shfl_xor(sum, 32);
    if (col) {
        float *dst = ell_val + ((int64_t)(n * 3 + i) * V + l31) * V;            // [n][i][w][v]
        const float *as = a_sum + (int64_t)i * V * V + l31;                       // [i][v][w]
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int v = (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (v < V) dst[v] = acc[r] / sum + as[v * V];
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
