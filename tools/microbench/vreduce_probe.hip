// Microbenchmark (round 6, review item 8): the V-dimension reduction of the graph conv -- agg[c][q] = sum over <= 6 skeleton
// neighbours of joint(q) -- formed per lane (a lane owns output column q = frame * V + joint, V = 25) from an x row staged in LDS:
//   GATHER   6 ds_read_b32 at per-lane offsets (what gcn_stage_sparse2_kernel / gcn16_kernel do)
//   SHUFFLE  1 ds_read_b32 of the lane's own column + 6 ds_bpermute_b32 (north_star's "wavefront shuffles"); the sources of a
//            frame that straddles the 64-lane wave (V = 25 does not divide 64) are not in the wave: those lanes fall back to reads
// Both do the same 6 FMAs per (channel, column).  Prints ns per 1000 (channel, column) aggregates per CU and the ratio.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
constexpr int V = 25, NCOL = 256, ROWS = 8;

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(float *out, const int *src, const float *val, int iters) {
    __shared__ float xs[ROWS][NCOL + 32];
    const int tid = threadIdx.x, lane = tid & 63, wbase = tid & ~63;
    for (int i = tid; i < ROWS * (NCOL + 32); i += 256) (&xs[0][0])[i] = (float)((i * 7 + blockIdx.x) % 13) * 0.01f;
    __syncthreads();
    const int q = tid, t = q / V, w = q - t * V;
    int off[6], lsrc[6];
    float ev[6];
    bool inwave = true;
    for (int e = 0; e < 6; ++e) {
        const int s = t * V + src[w * 6 + e];
        off[e] = s;
        ev[e] = val[w * 6 + e];
        lsrc[e] = (s - wbase) * 4;                         // byte index of the source lane for ds_bpermute
        inwave = inwave && s >= wbase && s < wbase + 64;
    }
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const float *row = xs[r];
            float a;
            if (MODE == 0) {
                a = ev[0] * row[off[0]];
#pragma unroll
                for (int e = 1; e < 6; ++e) a = fmaf(ev[e], row[off[e]], a);
            } else {
                const float own = row[q];
                if (inwave) {
                    a = ev[0] * __int_as_float(__builtin_amdgcn_ds_bpermute(lsrc[0], __float_as_int(own)));
#pragma unroll
                    for (int e = 1; e < 6; ++e) a = fmaf(ev[e], __int_as_float(__builtin_amdgcn_ds_bpermute(lsrc[e], __float_as_int(own))), a);
                } else {
                    a = ev[0] * row[off[0]];
#pragma unroll
                    for (int e = 1; e < 6; ++e) a = fmaf(ev[e], row[off[e]], a);
                }
            }
            acc += a;
        }
        asm volatile("" : "+v"(acc));
    }
    out[blockIdx.x * 256 + tid] = acc;
}

template <int MODE> double run(int blocks, int iters, const int *src, const float *val) {
    float *out; hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, src, val, iters);
    hipEventRecord(e0);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, src, val, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); hipFree(out);
    return ms / 5;
}
int main() {
    // NTU-like neighbour table: self, parent, up to 4 children (chain + a few branches), all inside the skeleton
    std::vector<int> src(V * 6); std::vector<float> val(V * 6);
    for (int w = 0; w < V; ++w)
        for (int e = 0; e < 6; ++e) { src[w * 6 + e] = e == 0 ? w : e == 1 ? (w + V - 1) % V : (w + e * 3) % V; val[w * 6 + e] = 0.1f * (e + 1); }
    int *dsrc; float *dval;
    hipMalloc(&dsrc, src.size() * 4); hipMalloc(&dval, val.size() * 4);
    hipMemcpy(dsrc, src.data(), src.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dval, val.data(), val.size() * 4, hipMemcpyHostToDevice);
    const int blocks = 512, iters = 4000;
    const double g = run<0>(blocks, iters, dsrc, dval), s = run<1>(blocks, iters, dsrc, dval);
    const double aggs = (double)blocks * 256 * ROWS * iters;
    printf("VREDUCE gather (6 ds_read_b32): %.3f ms = %.2f ps per aggregate | shuffle (1 read + 6 ds_bpermute, straddling frames read): %.3f ms = %.2f ps | shuffle / gather = %.3f\n",
           g, g * 1e9 / aggs, s, s * 1e9 / aggs, s / g);
    return 0;
}
