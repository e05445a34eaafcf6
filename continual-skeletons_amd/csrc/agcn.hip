// agcn.hip -- A-GCN adaptive adjacency (models/a_gcn/a_gcn.py:53-63) on gfx950.
#include "mfma_core.h"

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
                                                             int64_t e_seg_stride, int64_t e_chan_stride,
                                                             int seg_per_group, int64_t e_group_stride) {
    // logits = Ea^T . Eb over K = inter*T rows as an fp32-MFMA product: A[i = v][k] = Ea[row k][v],
    // B[k][j = w] = Eb[row k][w] (V <= 32 columns used), operands straight from global memory (each row is V
    // contiguous floats).  The four waves take interleaved k-steps (2 rows each) and their partial 32x32 tiles
    // are summed through LDS.
    __shared__ float part[4][32][33];
    const int n = blockIdx.x, i = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5;
    const int K = inter * T;
    const float *eseg = E + (int64_t)(n / seg_per_group) * e_group_stride + (int64_t)(n % seg_per_group) * e_seg_stride;
    const float *ea = eseg + (int64_t)i * inter * e_chan_stride;
    const float *eb = eseg + (int64_t)(3 + i) * inter * e_chan_stride;
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

extern "C" int csk_agcn_attention_f32(const float *E, const float *a_sum, float *ell_val, int n_seg, int inter, int T,
                                      int V, int64_t e_seg_stride, int64_t e_chan_stride, int seg_per_group,
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
    hipLaunchKernelGGL(agcn_attention_kernel, dim3(n_seg, 3), dim3(256), 0, (hipStream_t)stream, E, a_sum, ell_val,
                       inter, T, V, e_seg_stride, e_chan_stride, seg_per_group, e_group_stride);
    return (int)hipGetLastError();
}
