// head.hip -- the steps either side of the block stack: input permute + data_bn (models/st_gcn/st_gcn.py:49-57,
// models/base.py:73-82), spatial/temporal mean and FC (st_gcn.py:60-64).  HBM-bound helpers.
#include "mfma_core.h"

// ------------------------------------------------------------------------------------------------
// pre / post
// ------------------------------------------------------------------------------------------------
__global__ void input_norm_kernel(const float *__restrict__ x, const float *__restrict__ scale,
                                  const float *__restrict__ shift, float *__restrict__ h, int C, int T, int V, int M,
                                  int64_t h_seg_stride, int64_t h_chan_stride, int64_t total) {
    // one thread per input element, x index = (((n*C + c)*T + t)*V + v)*M + m  (reads fully coalesced)
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = i;
        const int m = r % M; r /= M;
        const int v = r % V; r /= V;
        const int t = r % T; r /= T;
        const int c = r % C;
        const int64_t n = r / C;
        const int ch = (m * V + v) * C + c;
        h[(n * M + m) * h_seg_stride + (int64_t)c * h_chan_stride + (int64_t)t * V + v] = fmaf(x[i], scale[ch], shift[ch]);
    }
}

// feat[n, c] = mean over m and over TV positions; one wave per (n, c)
__global__ __launch_bounds__(256) void pool_kernel(const float *__restrict__ h, float *__restrict__ feat, int N, int M,
                                                   int C, int TV, float scale) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;   // (n, c)
    if (row >= (int64_t)N * C) return;
    const int n = row / C, c = row % C;
    float s = 0.f;
    for (int m = 0; m < M; ++m) {
        const float *src = h + ((int64_t)(n * M + m) * C + c) * TV;
        // eight loads in flight per lane, added in the order of the plain loop (same sums, bit for bit): at batch 1 a launch
        // is 256 waves whose 59 dependent load round trips were the whole 20 us of the kernel
        int j = lane;
        for (; j + 7 * 64 < TV; j += 8 * 64) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[j + 64 * u];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; j < TV; j += 64) s += src[j];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) feat[row] = s / (float)((int64_t)M * TV) * scale;
}

// logits[n, k] = feat[n] . fc_w[k] + fc_b[k].  One THREAD per output, a wave = 64 classes of one sample: the feature row is
// wave-uniform (scalar loads), every lane streams its own weight row 16 bytes at a time (a 64-byte line serves four
// iterations), the sum runs in four interleaved chains over c -- no cross-lane reduction.  (First form: one wave per output with a 6-step
// butterfly each; 98 us per cycle for the 400 Kinetics classes of 4096 skeleton frames, profiles/r03d_coagcn_online_1shard.md.)
__global__ __launch_bounds__(256) void fc_kernel(const float *__restrict__ feat, const float *__restrict__ w,
                                                 const float *__restrict__ b, float *__restrict__ logits, int N, int C,
                                                 int classes, int ktiles) {
    const int lane = threadIdx.x & 63;
    const int64_t wid = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (wid >= (int64_t)N * ktiles) return;
    const int n = (int)(wid / ktiles), k = (int)(wid % ktiles) * 64 + lane;
    const float *wr = w + (int64_t)min(k, classes - 1) * C;
    const float *fr = feat + (int64_t)n * C;
    // four independent chains (channel c goes to chain c mod 4), combined pairwise: short dependency chains for the issue
    // rate and a summation error of the order of the tree the first form used
    f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
    int c = 0;
    if ((C & 3) == 0) {
        for (; c < C; c += 4) {
            const f32x4 wv = *reinterpret_cast<const f32x4 *>(wr + c);
            const f32x4 fv = *reinterpret_cast<const f32x4 *>(fr + c);
#pragma unroll
            for (int i = 0; i < 4; ++i) s4[i] = fmaf(fv[i], wv[i], s4[i]);
        }
    }
    for (; c < C; ++c) s4[c & 3] = fmaf(fr[c], wr[c], s4[c & 3]);
    const float s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    if (k < classes) logits[(int64_t)n * classes + k] = s + b[k];
}

// ------------------------------------------------------------------------------------------------
// Continual head in ONE launch (models/base.py:84-101): spatial_pool of a channel-major frame -> co.AvgPool1d window ->
// co.Linear.  One workgroup per stream n:
//   (1) feat[c] = mean of the MV = M*V positions of stream n in row c (one wave per row, the summation order of
//       co_spatial_pool_kernel: lane-strided partial sums, then the xor butterfly), written to ring slot `head`;
//   (2) if the window emits: pooled[c] = (sum of the `count` newest ring entries, oldest first) / window (co_window_mean_kernel's
//       order; the newest entry is taken from LDS -- same bits as the value just stored);
//   (3) logits[k] = pooled . fc_w[k] + fc_b[k] with fc_kernel's four interleaved chains.
// Bit-identical to the three separate launches (csk_co_spatial_pool_f32, csk_co_window_mean_f32, csk_fc_f32); at 1024
// streams it replaces 92 us of dependent launches by one of ~30 us and the pooled round trips by LDS.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void co_head_kernel(const float *__restrict__ h, float *__restrict__ ring, float *__restrict__ pooled,
                                                      const float *__restrict__ w, const float *__restrict__ b, float *__restrict__ logits,
                                                      int N, int C, int MV, int64_t P, int window, int head, int count, int emit,
                                                      int classes) {
    extern __shared__ __attribute__((aligned(16))) float hs[];              // feat [Cp] | pooled [Cp], Cp = C rounded up to 4
    const int Cp = (C + 3) & ~3;
    float *feat_s = hs, *pool_s = hs + Cp;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = blockIdx.x;
    const int64_t n_elem = (int64_t)N * C;
    if (h) {
        const float *src0 = h + (int64_t)n * MV;
        constexpr int RF = 16;                                              // rows of this wave in flight
        for (int c0 = wave; c0 < C; c0 += 4 * RF) {
            float s[RF];
#pragma unroll
            for (int u = 0; u < RF; ++u) {
                const int c = min(c0 + 4 * u, C - 1);
                const float *src = src0 + (int64_t)c * P;
                float a = 0.f;
                for (int j = lane; j < MV; j += 64) a += src[j];
                s[u] = a;
            }
#pragma unroll
            for (int u = 0; u < RF; ++u) {
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) s[u] += __shfl_xor(s[u], o);
                const int c = c0 + 4 * u;
                if (lane == 0 && c < C) {
                    const float f = s[u] / (float)MV;
                    feat_s[c] = f;
                    ring[(int64_t)head * n_elem + (int64_t)n * C + c] = f;
                }
            }
        }
    } else {                                                                 // end padding of the window: a zero feature
        for (int c = tid; c < C; c += 256) {
            feat_s[c] = 0.f;
            ring[(int64_t)head * n_elem + (int64_t)n * C + c] = 0.f;
        }
    }
    if (!emit) return;
    __syncthreads();
    for (int c = tid; c < Cp; c += 256) {
        float s = 0.f;
        if (c < C) {
            // eight ring entries in flight per thread (a rolled loop waits for every load: 75 dependent round trips);
            // the additions stay in oldest-first order
            for (int j0 = count - 1; j0 >= 1; j0 -= 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    int slot = (head - max(j0 - u, 1)) % window;
                    if (slot < 0) slot += window;
                    v[u] = ring[(int64_t)slot * n_elem + (int64_t)n * C + c];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (j0 - u >= 1) s += v[u];
            }
            if (count >= 1) s += feat_s[c];
            s = s / (float)window;
            pooled[(int64_t)n * C + c] = s;
        }
        pool_s[c] = s;
    }
    __syncthreads();
    // FC: fc_kernel's four interleaved chains (channel c goes to chain c mod 4, combined as (s0 + s1) + (s2 + s3)), one chain
    // per THREAD here -- four neighbouring lanes share a class -- so that 256 threads work on 64 classes at a time instead of
    // 60 threads walking 256 channels each (one stream: a single workgroup does the whole head).  Same sums, same order.
    const int chain = tid & 3;
    for (int k0 = 0; k0 < classes; k0 += 64) {
        const int k = k0 + (tid >> 2);
        const float *wr = w + (int64_t)min(k, classes - 1) * C;
        float sc = 0.f;
        int c = chain;
        for (; c + 28 < C; c += 32) {                       // eight weight loads in flight
            float wv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) wv[u] = wr[c + 4 * u];
#pragma unroll
            for (int u = 0; u < 8; ++u) sc = fmaf(pool_s[c + 4 * u], wv[u], sc);
        }
        for (; c < C; c += 4) sc = fmaf(pool_s[c], wr[c], sc);
        sc += __shfl_xor(sc, 1);                            // lanes 0 / 2 of a quad: s0 + s1, s2 + s3
        sc += __shfl_xor(sc, 2);                            // (s0 + s1) + (s2 + s3)
        if (chain == 0 && k < classes) logits[(int64_t)n * classes + k] = sc + b[k];
    }
}

extern "C" int csk_co_head_step_f32(const float *h, float *pool_ring, float *pooled, const float *fc_w, const float *fc_b,
                                    float *logits, int N, int C, int MV, int64_t P, int window, int head, int count, int emit,
                                    int classes, void *stream) {
    if (!pool_ring) CSK_FAIL("co_head_step: null pointer");
    if (N <= 0 || C <= 0 || C > 8192 || MV <= 0 || (h && (int64_t)N * MV > P) || window <= 0 || head < 0 || head >= window || count < 0 ||
        count > window)
        CSK_FAIL("co_head_step: bad dims");
    if (emit && (!pooled || !fc_w || !fc_b || !logits || classes <= 0 || count < 1)) CSK_FAIL("co_head_step: an emitting step needs pooled, fc_w, fc_b, logits and count >= 1");
    const size_t lds = 2 * (size_t)((C + 3) & ~3) * sizeof(float);
    hipLaunchKernelGGL(co_head_kernel, dim3((unsigned)N), dim3(256), lds, (hipStream_t)stream, h, pool_ring, pooled, fc_w, fc_b, logits, N,
                       C, MV, P, window, head, count, emit, classes);
    return (int)hipGetLastError();
}

// input permute + data_bn of up to 8 frames of a launch cycle in ONE launch (continual form: T = 1 per frame):
// frame f: x = src[f] (N, C, V, M) -> dst[f] channel-major (C, P)
struct NormFrames {
    const float *src[8];
    float *dst[8];
};
__global__ void input_norm_frames_kernel(const NormFrames f, const float *__restrict__ scale, const float *__restrict__ shift, int C,
                                         int V, int M, int64_t P, int64_t total) {
    const float *x = f.src[blockIdx.y];
    float *h = f.dst[blockIdx.y];
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = i;
        const int m = r % M; r /= M;
        const int v = r % V; r /= V;
        const int c = r % C;
        const int64_t n = r / C;
        const int ch = (m * V + v) * C + c;
        h[(n * M + m) * V + (int64_t)c * P + v] = fmaf(x[i], scale[ch], shift[ch]);
    }
}

extern "C" int csk_input_norm_frames_f32(const float *const *frames, float *const *dst, int r, const float *scale, const float *shift,
                                         int N, int C, int V, int M, int64_t P, void *stream) {
    if (!frames || !dst || !scale || !shift) CSK_FAIL("input_norm_frames: null pointer");
    if (r < 1 || r > 8 || N <= 0 || C <= 0 || V <= 0 || M <= 0 || P < (int64_t)N * M * V) CSK_FAIL("input_norm_frames: bad dims (1..8 frames)");
    NormFrames f;
    for (int i = 0; i < 8; ++i) {
        f.src[i] = frames[i < r ? i : r - 1];
        f.dst[i] = dst[i < r ? i : r - 1];
        if (!f.src[i] || !f.dst[i]) CSK_FAIL("input_norm_frames: null frame");
    }
    const int64_t total = (int64_t)N * C * V * M;
    const int64_t want = (total + 255) / 256;
    const int blocks = (int)(want < 4096 ? want : 4096);
    hipLaunchKernelGGL(input_norm_frames_kernel, dim3(blocks, r), dim3(256), 0, (hipStream_t)stream, f, scale, shift, C, V, M, P, total);
    return (int)hipGetLastError();
}

extern "C" int csk_input_norm_f32(const float *x, const float *scale, const float *shift, float *h, int N, int C,
                                  int T, int V, int M, int64_t h_seg_stride, int64_t h_chan_stride, void *stream) {
    if (!x || !scale || !shift || !h) CSK_FAIL("input_norm: null pointer");
    if (N <= 0 || C <= 0 || T <= 0 || V <= 0 || M <= 0) CSK_FAIL("input_norm: bad dims");
    const int64_t total = (int64_t)N * C * T * V * M;
    const int64_t want = (total + 255) / 256;
    const int blocks = (int)(want < 8192 ? want : 8192);
    hipLaunchKernelGGL(input_norm_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, scale, shift, h, C, T, V,
                       M, h_seg_stride, h_chan_stride, total);
    return (int)hipGetLastError();
}

extern "C" int csk_fc_f32(const float *feat, const float *fc_w, const float *fc_b, float *logits, int N, int C,
                          int classes, void *stream) {
    if (!feat || !fc_w || !fc_b || !logits) CSK_FAIL("fc: null pointer");
    if (N <= 0 || C <= 0 || classes <= 0) CSK_FAIL("fc: bad dims");
    // the 16-byte loads of the C % 4 == 0 path need 16-byte aligned rows; other channel counts take the scalar loop
    if ((C & 3) == 0 && ((reinterpret_cast<uintptr_t>(feat) & 15) || (reinterpret_cast<uintptr_t>(fc_w) & 15)))
        CSK_FAIL("fc: feat / fc_w must be 16-byte aligned when C is a multiple of 4");
    const int ktiles = (classes + 63) / 64;
    const int64_t waves = (int64_t)N * ktiles;
    hipLaunchKernelGGL(fc_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, feat, fc_w, fc_b,
                       logits, N, C, classes, ktiles);
    return (int)hipGetLastError();
}

extern "C" int csk_pool_fc_f32(const float *h, const float *fc_w, const float *fc_b, float *feat, float *logits, int N,
                               int M, int C, int TV, int classes, void *stream) {
    if (!h || !feat) CSK_FAIL("pool_fc: null pointer");
    if (N <= 0 || M <= 0 || C <= 0 || TV <= 0) CSK_FAIL("pool_fc: bad dims");
    const int64_t rows = (int64_t)N * C;
    hipLaunchKernelGGL(pool_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, h, feat, N, M,
                       C, TV, 1.0f);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    if (logits) return csk_fc_f32(feat, fc_w, fc_b, logits, N, C, classes, stream);
    return 0;
}

extern "C" int csk_pool_scaled_f32(const float *h, float *feat, int N, int M, int C, int TV, float scale, void *stream) {
    if (!h || !feat) CSK_FAIL("pool_scaled: null pointer");
    if (N <= 0 || M <= 0 || C <= 0 || TV <= 0) CSK_FAIL("pool_scaled: bad dims");
    const int64_t rows = (int64_t)N * C;
    hipLaunchKernelGGL(pool_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, h, feat, N, M,
                       C, TV, scale);
    return (int)hipGetLastError();
}

