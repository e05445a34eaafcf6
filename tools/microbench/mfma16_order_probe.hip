// Is v_mfma_f32_16x16x4_f32 the same fmaf chain (ascending k) as two v_mfma_f32_32x32x2_f32 and as scalar fmaf?
// Prints the number of bitwise mismatches over random operands (0 expected for "chain" forms).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// A [16][K] row-major, B [K][16]; D16 [16][16] via 16x16x4 over K in steps of 4
__global__ void k16(const float *A, const float *B, float *D, int K) {
    const int l = threadIdx.x, n = l & 15, k = l >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[n * K + k0 + k], B[(k0 + k) * 16 + n], acc, 0, 0, 0);
    for (int g = 0; g < 4; ++g) D[(4 * k + g) * 16 + n] = acc[g];
}
// the same product with 32x32x2 (rows/cols 16..31 zero)
__global__ void k32(const float *A, const float *B, float *D, int K) {
    const int l = threadIdx.x, n = l & 31, k = l >> 5;
    f32x16 acc;
    for (int g = 0; g < 16; ++g) acc[g] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 2) {
        const float a = n < 16 ? A[n * K + k0 + k] : 0.f, b = n < 16 ? B[(k0 + k) * 16 + n] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    // D row = 8*(g>>2) + 4*k... 32x32 layout: lane (n, kh): rows (g&3) + 8*(g>>2) + 4*kh, col n
    for (int g = 0; g < 16; ++g) {
        const int row = (g & 3) + 8 * (g >> 2) + 4 * k;
        if (row < 16 && n < 16) D[row * 16 + n] = acc[g];
    }
}
int main() {
    const int K = 64;
    std::vector<float> A(16 * K), B(K * 16), ref(256), d16(256), d32(256);
    srand(1);
    for (auto &v : A) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    for (auto &v : B) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            float s = 0.f;
            for (int k = 0; k < K; ++k) s = fmaf(A[i * K + k], B[k * 16 + j], s);
            ref[i * 16 + j] = s;
        }
    float *dA, *dB, *dD;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, 256 * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, dA, dB, dD, K);
    hipMemcpy(d16.data(), dD, 1024, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, dA, dB, dD, K);
    hipMemcpy(d32.data(), dD, 1024, hipMemcpyDeviceToHost);
    int m16 = 0, m32 = 0, m1632 = 0;
    for (int i = 0; i < 256; ++i) {
        m16 += memcmp(&d16[i], &ref[i], 4) != 0;
        m32 += memcmp(&d32[i], &ref[i], 4) != 0;
        m1632 += memcmp(&d16[i], &d32[i], 4) != 0;
    }
    printf("MFMA16_ORDER mismatches vs fmaf chain: 16x16x4 %d, 32x32x2 %d; 16x16x4 vs 32x32x2 %d (of 256)\n", m16, m32, m1632);
    return 0;
}
