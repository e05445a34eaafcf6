// Microbenchmark: do the two workgroups that share a CU run the K loop of the stage kernels in LOCKSTEP (both at their
// barrier / LDS commit at the same time, the matrix pipe idle meanwhile), and what does breaking the symmetry buy?
// The loop is `struct 3` of mfma_loop_probe.hip (the per-chunk structure of tcn_stage_kernel at 128 x 128: two barriers,
// LDS commit of 9 f32x4 + 18 dwords, 27 prefetch loads in three bursts, 144 MFMAs per wave and chunk).
//   MODE 0  as shipped (s_setprio 1 around the MFMA segments of every wave)
//   MODE 1  stagger: the workgroup in the ODD wave slot of its SIMDs (HW_ID.wave_id of its first wave) sleeps `delay`
//           x 64 cycles before its first chunk
//   MODE 2  static priority: odd-slot workgroup runs at s_setprio 2 throughout, even-slot at 0, no per-segment flips
//   MODE 3  both
//   MODE 4  no s_setprio at all (reference for what the per-segment flips are worth)
// hipcc -O3 --offload-arch=gfx950 desync_probe.hip -o bin/desync_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int MT = 128, KC = 8, LDB = 376, TAPS = 9;

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(float *out, int *slot_out, const float *gw, const float *gb, int chunks, int delay) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ int slot_s;
    float *Wl = smem, *Bl = smem + TAPS * KC * MT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, kh = lane >> 5;
    for (int i = tid; i < TAPS * KC * MT + KC * LDB; i += 256) smem[i] = (float)((i * 7 + blockIdx.x) % 13) * 0.01f;
    if (tid == 0) slot_s = (int)__builtin_amdgcn_s_getreg(6148) & 15;        // HW_REG_HW_ID bits [3:0]: wave slot on the SIMD
    __syncthreads();
    const int odd = __builtin_amdgcn_readfirstlane(slot_s) & 1;
    if (tid == 0) slot_out[blockIdx.x] = slot_s;
    const int offA = (wave & 1) * 64 + l31, off0 = (wave >> 1) * 64 + l31, off1 = off0 + 32;
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int g = 0; g < 16; ++g) acc[a][b][g] = 0.f;
    f32x4 wv[9];
    float bv[18];
    for (int u = 0; u < 9; ++u) wv[u] = *reinterpret_cast<const f32x4 *>(gw + (u * 256 + tid) * 4);
    for (int u = 0; u < 18; ++u) bv[u] = gb[(size_t)blockIdx.x * 65536 + u * 256 + tid];
    if (MODE == 1 || MODE == 3) {
        if (odd) for (int i = 0; i < delay; ++i) __builtin_amdgcn_s_sleep(1);
    }
    if (MODE == 2 || MODE == 3) {
        if (odd) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0);
    }
    for (int c = 0; c < chunks; ++c) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 9; ++u) *reinterpret_cast<f32x4 *>(Wl + (u * 256 + tid) * 4) = wv[u];
#pragma unroll
        for (int u = 0; u < 18; ++u) Bl[(u >> 1) % KC * LDB + (u & 1) * 128 + (tid & 127)] = bv[u] + (float)(tid >> 7);
        __syncthreads();
        const float *gwc = gw + (size_t)((c + 1) & 31) * 9216, *gbc = gb + (size_t)blockIdx.x * 65536 + (size_t)((c + 1) & 7) * 4608;
        if (MODE == 5 || MODE == 6) {
            // trickle: the 27 loads spread one by one over the 36 k-steps of the chunk (fully unrolled), pinned with
            // sched_group_barrier (MODE 5: 4 DS reads, 4 MFMAs, <= 1 VMEM per k-step) or left to the compiler (MODE 6)
#pragma unroll
            for (int i = 0; i < 36; ++i) {
                const int r = i / 4, s = i % 4;
                const float *wr = Wl + r * (KC * MT) + offA + kh * MT, *br = Bl + r * 25 + kh * LDB;
                const float a0 = wr[2*s*MT], a1 = wr[2*s*MT+32], b0 = br[2*s*LDB+off0], b1 = br[2*s*LDB+off1];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
                if (i < 27) {
                    if (i % 3 == 0) wv[i / 3] = *reinterpret_cast<const f32x4 *>(gwc + ((i / 3) * 256 + tid) * 4);
                    else { const int u = 2 * (i / 3) + (i % 3 - 1); bv[u] = gbc[u * 256 + tid]; }
                }
                if (MODE == 5) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                    if (i < 27) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
            }
            continue;
        }
#pragma unroll
        for (int g3 = 0; g3 < 3; ++g3) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int u = 3 * g3 + j;
                wv[u] = *reinterpret_cast<const f32x4 *>(gwc + (u * 256 + tid) * 4);
                bv[2 * u] = gbc[2 * u * 256 + tid];
                bv[2 * u + 1] = gbc[(2 * u + 1) * 256 + tid];
            }
            if (MODE == 0 || MODE == 1) __builtin_amdgcn_s_setprio(1);
            for (int r = 3 * g3; r < 3 * g3 + 3; ++r) {
                const float *wr = Wl + r * (KC * MT) + offA + kh * MT, *br = Bl + r * 25 + kh * LDB;
#pragma unroll
                for (int s = 0; s < KC / 2; ++s) {
                    const float a0 = wr[2*s*MT], a1 = wr[2*s*MT+32], b0 = br[2*s*LDB+off0], b1 = br[2*s*LDB+off1];
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
                }
            }
            if (MODE == 0 || MODE == 1) __builtin_amdgcn_s_setprio(0);
        }
    }
    float s = 0.f;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int g = 0; g < 16; ++g) s += acc[a][b][g];
    for (int u = 0; u < 9; ++u) s += wv[u][0];
    for (int u = 0; u < 18; ++u) s += bv[u];
    out[blockIdx.x * 256 + tid] = s;
}

template <int MODE> double run(int chunks, int blocks, int delay, bool census = false) {
    float *out, *gw, *gb;
    int *slot;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipMalloc(&slot, (size_t)blocks * 4);
    hipMalloc(&gw, (size_t)32 * 9216 * 4 + 65536); hipMemset(gw, 0, (size_t)32 * 9216 * 4 + 65536);
    hipMalloc(&gb, (size_t)blocks * 65536 * 4 + (1 << 20)); hipMemset(gb, 0, (size_t)blocks * 65536 * 4 + (1 << 20));
    const size_t lds = (TAPS * KC * MT + KC * LDB) * 4;
    hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), lds, 0, out, slot, gw, gb, chunks, delay);
    hipEventRecord(e0);
    for (int it = 0; it < 5; ++it) hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), lds, 0, out, slot, gw, gb, chunks, delay);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    if (census) {
        std::vector<int> h(blocks);
        hipMemcpy(h.data(), slot, (size_t)blocks * 4, hipMemcpyDeviceToHost);
        int cnt[16] = {0};
        for (int v : h) cnt[v & 15]++;
        printf("wave-slot census over %d workgroups:", blocks);
        for (int i = 0; i < 16; ++i) if (cnt[i]) printf(" slot %d: %d", i, cnt[i]);
        printf("\n");
    }
    hipFree(out); hipFree(gw); hipFree(gb); hipFree(slot);
    const double flops = (double)blocks * 4 * chunks * TAPS * (KC / 2) * 4 * 4096.0;
    return flops / (ms * 1e-3) / 1e12;
}

int main() {
    // chunks: 16 = a C = 128 layer's K loop, 32 = C = 256; blocks: 512 = one round, 800 = the online launches, 3072 = six rounds
    const int shapes[][2] = {{64, 3072}, {16, 3072}, {16, 512}, {16, 800}, {32, 800}};
    for (auto &sh : shapes) {
        const int chunks = sh[0], blocks = sh[1];
        printf("== chunks %d, workgroups %d\n", chunks, blocks);
        printf("mode 0 (as shipped)                 : %.1f TFLOP/s\n", run<0>(chunks, blocks, 0, true));
        printf("mode 4 (no setprio)                 : %.1f TFLOP/s\n", run<4>(chunks, blocks, 0));
        for (int d : {36, 72, 144, 288})
            printf("mode 1 (stagger %4d x 64 cycles)   : %.1f TFLOP/s\n", d, run<1>(chunks, blocks, d));
        printf("mode 2 (static prio by slot parity) : %.1f TFLOP/s\n", run<2>(chunks, blocks, 0));
        for (int d : {72, 144})
            printf("mode 3 (prio + stagger %4d)        : %.1f TFLOP/s\n", d, run<3>(chunks, blocks, d));
        printf("mode 5 (loads trickled, pinned)     : %.1f TFLOP/s\n", run<5>(chunks, blocks, 0));
        printf("mode 6 (loads interleaved, compiler): %.1f TFLOP/s\n", run<6>(chunks, blocks, 0));
        printf("mode 0 again                        : %.1f TFLOP/s\n", run<0>(chunks, blocks, 0));
    }
    return 0;
}
