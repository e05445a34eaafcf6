// Microbenchmark: can the fp32 VECTOR pipe do GEMM work beside the fp32 MATRIX pipe?  On gfx950 both run fp32 FMAs at
// 64 FLOP/clk/SIMD (MI355X_MICROARCH.md), and an MFMA holds the SIMD's vector issue only for a few of its 64 cycles.
// Loop = `struct 3` of mfma_loop_probe.hip (128 x 128 MFMA tile, two barriers, LDS commit, 27 prefetch loads per chunk) plus
// a VALU side tile: wave w accumulates RV rows x 64 columns (lane = column) with the weights as wave-uniform SCALAR operands
// (s_load from the packed weights) and the activation value from the LDS tile the MFMA side stages anyway:
//     vacc[i] = fmaf(W[tap][ch][wave * RV + i]  (SGPR),  B[tap][ch][128 + lane]  (VGPR <- ds_read),  vacc[i])
// i.e. per (tap, channel) one ds_read_b32 + RV v_fma_f32 per wave next to 2 MFMAs.
//   RV =  0  MFMA only
//   RV = 16  VALU tile  64 rows x 64 columns per workgroup (+12.5 % work)
//   RV = 32  VALU tile 128 rows x 64 columns per workgroup (+50 % work on 192 columns)
// hipcc -O3 --offload-arch=gfx950 hybrid_probe.hip -o bin/hybrid_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int MT = 128, KC = 8, LDB = 376, TAPS = 9;

template <int RV, int PIN>
__global__ __launch_bounds__(256, 2) void k(float *out, const float *__restrict__ gw, const float *__restrict__ gb,
                                            const float *__restrict__ gws, int chunks) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Wl = smem, *Bl = smem + TAPS * KC * MT;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, kh = lane >> 5;
    for (int i = tid; i < TAPS * KC * MT + KC * LDB; i += 256) smem[i] = (float)((i * 7 + blockIdx.x) % 13) * 0.01f;
    __syncthreads();
    const int offA = (wave & 1) * 64 + l31, off0 = (wave >> 1) * 64 + l31, off1 = off0 + 32;
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int g = 0; g < 16; ++g) acc[a][b][g] = 0.f;
    float vacc[RV > 0 ? RV : 1];
    for (int i = 0; i < (RV > 0 ? RV : 1); ++i) vacc[i] = 0.f;
    float winv[RV > 0 ? RV : 1];
    for (int i = 0; i < (RV > 0 ? RV : 1); ++i) winv[i] = gws[wave * 32 + i];
    f32x4 wv[9];
    float bv[18];
    for (int u = 0; u < 9; ++u) wv[u] = *reinterpret_cast<const f32x4 *>(gw + (u * 256 + tid) * 4);
    for (int u = 0; u < 18; ++u) bv[u] = gb[(size_t)blockIdx.x * 65536 + u * 256 + tid];
    for (int c = 0; c < chunks; ++c) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 9; ++u) *reinterpret_cast<f32x4 *>(Wl + (u * 256 + tid) * 4) = wv[u];
#pragma unroll
        for (int u = 0; u < 18; ++u) Bl[(u >> 1) % KC * LDB + (u & 1) * 128 + (tid & 127)] = bv[u] + (float)(tid >> 7);
        __syncthreads();
        const float *gwc = gw + (size_t)((c + 1) & 31) * 9216, *gbc = gb + (size_t)blockIdx.x * 65536 + (size_t)((c + 1) & 7) * 4608;
        const float *wsc = gws + (size_t)(c & 31) * 9216 + wave * (RV > 0 ? RV : 1);      // this chunk's weights [tap][ch][128 rows]
#pragma unroll
        for (int g3 = 0; g3 < 3; ++g3) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int u = 3 * g3 + j;
                wv[u] = *reinterpret_cast<const f32x4 *>(gwc + (u * 256 + tid) * 4);
                bv[2 * u] = gbc[2 * u * 256 + tid];
                bv[2 * u + 1] = gbc[(2 * u + 1) * 256 + tid];
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int rr = 0; rr < 3; ++rr) {
                const int r = 3 * g3 + rr;
                const float *wr = Wl + r * (KC * MT) + offA + kh * MT, *br = Bl + r * 25 + kh * LDB;
#pragma unroll
                for (int s = 0; s < KC / 2; ++s) {
                    const float a0 = wr[2*s*MT], a1 = wr[2*s*MT+32], b0 = br[2*s*LDB+off0], b1 = br[2*s*LDB+off1];
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
                    if (RV > 0 && PIN == 2) {
#pragma unroll
                        for (int h = 0; h < 2; ++h)
#pragma unroll
                            for (int i = 0; i < RV; ++i) vacc[i] = __builtin_fmaf(winv[i], h ? b1 : b0, vacc[i]);
                    } else if (RV > 0) {
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int kk = 2 * s + h;
                            const float bvv = Bl[r * 25 + kk * LDB + 128 + lane];
                            const float *wrow = wsc + (r * KC + kk) * MT;                  // wave-uniform address -> s_load
#pragma unroll
                            for (int i = 0; i < RV; ++i) vacc[i] = __builtin_fmaf(wrow[i], bvv, vacc[i]);
                        }
                        if (PIN == 1) {         // per k-step: 4 MFMAs, each followed by a quarter of the 2 RV FMAs
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                                __builtin_amdgcn_sched_group_barrier(0x002, RV / 2, 0);
                            }
                        }
                    }
                }
            }
            __builtin_amdgcn_s_setprio(0);
        }
    }
    float s = 0.f;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int g = 0; g < 16; ++g) s += acc[a][b][g];
    for (int i = 0; i < (RV > 0 ? RV : 1); ++i) s += vacc[i];
    for (int u = 0; u < 9; ++u) s += wv[u][0];
    for (int u = 0; u < 18; ++u) s += bv[u];
    out[blockIdx.x * 256 + tid] = s;
}

template <int RV, int PIN> void run(int chunks, int blocks) {
    float *out, *gw, *gb, *gws;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipMalloc(&gw, (size_t)33 * 9216 * 4 + 65536); hipMemset(gw, 0, (size_t)33 * 9216 * 4 + 65536);
    hipMalloc(&gws, (size_t)33 * 9216 * 4 + 65536);
    {
        std::vector<float> h(33 * 9216 + 16384);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 31) % 17) * 0.01f - 0.05f;
        hipMemcpy(gws, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    }
    hipMalloc(&gb, (size_t)blocks * 65536 * 4 + (1 << 20)); hipMemset(gb, 0, (size_t)blocks * 65536 * 4 + (1 << 20));
    const size_t lds = (TAPS * KC * MT + KC * LDB) * 4;
    hipFuncSetAttribute((const void *)k<RV, PIN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<RV, PIN>), dim3(blocks), dim3(256), lds, 0, out, gw, gb, gws, chunks);
    hipEventRecord(e0);
    for (int it = 0; it < 5; ++it) hipLaunchKernelGGL((k<RV, PIN>), dim3(blocks), dim3(256), lds, 0, out, gw, gb, gws, chunks);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    hipFree(out); hipFree(gw); hipFree(gb); hipFree(gws);
    const double mf = (double)blocks * 4 * chunks * TAPS * (KC / 2) * 4 * 4096.0;
    const double vf = (double)blocks * 4 * chunks * TAPS * KC * RV * 64 * 2.0;
    printf("RV %2d pin %d: %.3f ms  MFMA %.1f + VALU %.1f = %.1f TFLOP/s\n", RV, PIN, ms, mf / (ms * 1e-3) / 1e12,
           vf / (ms * 1e-3) / 1e12, (mf + vf) / (ms * 1e-3) / 1e12);
}

int main() {
    const int chunks = 32, blocks = 512 * 6;
    run<0, 0>(chunks, blocks);
    run<8, 0>(chunks, blocks);
    run<16, 0>(chunks, blocks);
    run<32, 0>(chunks, blocks);
    run<8, 1>(chunks, blocks);
    run<16, 1>(chunks, blocks);
    run<32, 1>(chunks, blocks);
    run<0, 0>(chunks, blocks);
    run<8, 2>(chunks, blocks);
    run<16, 2>(chunks, blocks);
    run<32, 2>(chunks, blocks);
    return 0;
}
