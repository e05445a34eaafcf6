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
        for (int j = lane; j < TV; j += 64) s += src[j];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) feat[row] = s / (float)((int64_t)M * TV) * scale;
}

// logits[n, k] = feat[n] . fc_w[k] + fc_b[k]; one wave per output
__global__ __launch_bounds__(256) void fc_kernel(const float *__restrict__ feat, const float *__restrict__ w,
                                                 const float *__restrict__ b, float *__restrict__ logits, int N, int C,
                                                 int classes) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t o = (int64_t)blockIdx.x * 4 + wave;
    if (o >= (int64_t)N * classes) return;
    const int n = o / classes, k = o % classes;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s = fmaf(feat[(int64_t)n * C + c], w[(int64_t)k * C + c], s);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) logits[o] = s + b[k];
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
    const int64_t outs = (int64_t)N * classes;
    hipLaunchKernelGGL(fc_kernel, dim3((unsigned)((outs + 3) / 4)), dim3(256), 0, (hipStream_t)stream, feat, fc_w, fc_b,
                       logits, N, C, classes);
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

