// Microbenchmark: what would split-bf16 arithmetic buy on the MFMA side?  An fp32 product a*b is emulated by six bf16
// products of 3-way splits (hh, hm, mh, hl, lh, mm; tests/studies/bf16_split_study.py: 4.8e-6 on activations), issued as
// six v_mfma_f32_32x32x16_bf16 per 32x32 block and 16-deep K step, against the exact v_mfma_f32_32x32x2_f32 loop of the
// shipped kernels (eight of them per 16-deep K step).  Wave tile 64x64 (2x2 blocks), 256 threads, 2 workgroups per CU.
//   variant 0: fp32 MFMA, operands in registers            (the ceiling the stage kernels are priced against)
//   variant 1: 6 x bf16 MFMA, operands in registers        (MFMA-side ceiling of the split)
//   variant 2: 6 x bf16 MFMA, operands read from LDS per K step (3 pieces x 2 row blocks + 3 pieces x 2 column
//              blocks = 12 ds_read_b128 per 24 MFMAs) -- no staging, no splitting work: an upper bound of a real kernel
// Reports "fp32-equivalent" TFLOP/s = 2 * M * N * K / time.   hipcc --offload-arch=gfx950 -O3 -o bf16_split_probe ...
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int VARIANT>
__global__ __launch_bounds__(256, 2) void probe(float *out, int ksteps) {
    __shared__ __attribute__((aligned(16))) bf16x8 lds[3 * 2 * 64 * 2];     // [piece][A|B][lane][block]
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 3 * 2 * 64 * 2; i += 256)
        for (int e = 0; e < 8; ++e) lds[i][e] = (__bf16)(0.001f * (float)((i + e) % 17));
    __syncthreads();
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int g = 0; g < 16; ++g) acc[a][b][g] = 0.f;
    float fa0 = 0.001f * lane, fa1 = fa0 + 1.f, fb0 = 0.002f * lane, fb1 = fb0 + 1.f;
    bf16x8 pa[3][2], pb[3][2];
    for (int p = 0; p < 3; ++p) for (int b = 0; b < 2; ++b) { pa[p][b] = lds[((p * 2 + 0) * 64 + lane) * 2 + b]; pb[p][b] = lds[((p * 2 + 1) * 64 + lane) * 2 + b]; }
    for (int k = 0; k < ksteps; ++k) {
        if (VARIANT == 0) {
#pragma unroll
            for (int s = 0; s < 8; ++s) {           // 16-deep K step = 8 x (32x32x2)
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0, fb0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0, fb1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1, fb0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1, fb1, acc[1][1], 0, 0, 0);
                asm volatile("" : "+v"(fa0), "+v"(fb0));
            }
        } else {
            if (VARIANT == 2) {
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        pa[p][b] = lds[((p * 2 + 0) * 64 + lane) * 2 + ((b + k) & 1)];
                        pb[p][b] = lds[((p * 2 + 1) * 64 + lane) * 2 + ((b + k) & 1)];
                    }
            }
            // products hh, hm, mh, hl, lh, mm  (piece 0 = high, 1 = middle, 2 = low)
            constexpr int PA[6] = {0, 0, 1, 0, 2, 1}, PB[6] = {0, 1, 0, 2, 0, 1};
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[PA[t]][a], pb[PB[t]][b], acc[a][b], 0, 0, 0);
            if (VARIANT == 1) asm volatile("" : "+v"(pa[0][0]), "+v"(pb[0][0]));
        }
    }
    float s = 0.f;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int g = 0; g < 16; ++g) s += acc[a][b][g];
    out[blockIdx.x * 256 + tid] = s;
}

template <int V>
static void run(const char *name, float *out) {
    const int blocks = 512 * 8, ksteps = 4096;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<V><<<blocks, 256>>>(out, 64);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<V><<<blocks, 256>>>(out, ksteps);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = 2.0 * 64 * 64 * 16 * 4.0 * blocks * ksteps;     // per wave 64x64x16 MAC per K step, 4 waves
    printf("%-58s %8.3f ms  %7.1f fp32-equivalent TFLOP/s\n", name, ms, flop / ms / 1e9);
}

int main() {
    float *out;
    hipMalloc(&out, 512 * 8 * 256 * sizeof(float));
    run<0>("fp32 MFMA 32x32x2, register operands", out);
    run<1>("6 x bf16 MFMA 32x32x16 (3-way split), register operands", out);
    run<2>("6 x bf16 MFMA 32x32x16, 12 ds_read_b128 per K step", out);
    return 0;
}
