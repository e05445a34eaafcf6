// Microbenchmark (round 5, review item 2): the fp32 K-loop structure of tcn_stage_kernel with a DEDICATED LOADER WAVE.
// Compute waves issue only ds_read + MFMA (+ one synchronisation per chunk); one extra wave per workgroup streams the
// weight chunk and the activation tile into a 3-slot LDS ring with global_load_lds_dwordx4 (no staging VGPRs, no ds_write
// in the compute waves, no global loads among the MFMAs).  Same work and traffic per 8 channels as struct 3 of
// mfma_loop_probe.hip (the shipped structure: 138.5 TFLOP/s; barriers only: 150.2; plain loop: 153.1).
//   SYNC 0: one s_barrier per chunk shared by all waves (the loader waits for its DMAs in front of it)
//   SYNC 1: FULL / FREE words in LDS, no barrier at all (compute waves drift independently)
//   shape A: 128 x 128 tile, 4 compute waves + 1 loader, 4-channel half-chunks (24 KiB slots), 2 workgroups per CU
//   shape C: 128 x 256 tile, 8 compute waves + NL loaders, 8-channel chunks (52 KiB slots), 1 workgroup per CU
// hipcc --offload-arch=gfx950 -O3 loader_wave_probe.hip -o bin/loader_wave_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void;
constexpr int MT = 128, TAPS = 9;

// (ROCm 7.2: the builtin inside a kernel TEMPLATE body drops the host stub -- keep it in a plain device function)
__device__ __forceinline__ void dma16(const float *g, float *l) {
    __builtin_amdgcn_global_load_lds(g, (lds_void *)l, 16, 0, 0);
}
// LDS flag accesses of the LOADER wave as inline assembly: the compiler orders every LDS access it knows about behind ALL
// outstanding LDS-DMAs (s_waitcnt vmcnt(0)), which would serialise the ring; the flags never alias a DMA target.
__device__ __forceinline__ unsigned lds_peek(const volatile unsigned *p) {
    unsigned v;
    const unsigned a = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned *)p;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    return v;
}
__device__ __forceinline__ void lds_bump(volatile unsigned *p) {
    const unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned *)p, one = 1;
    asm volatile("ds_add_u32 %0, %1" ::"v"(a), "v"(one) : "memory");
}

template <int NCW, int NLW, int HC, int LDB, int SYNC, int PRIO, int LPRIO = 0, int MISAL = 0, int HBM = 0, int REG = 0>
__global__ __launch_bounds__((NCW + NLW) * 64) void lw_kernel(float *out, const float *gw, const float *gb, int nchunks) {
    constexpr int NS = 3;
    constexpr int WSZ = TAPS * HC * MT;                       // floats of weights per chunk
    constexpr int BSZ = (HC * LDB + 255) / 256 * 256;         // activation tile, padded to whole 1 KiB pieces
    constexpr int SLOT = WSZ + BSZ, PIECES = SLOT / 256;
    constexpr int PPW = (PIECES + NLW - 1) / NLW;             // pieces per loader wave
    static_assert(WSZ % 256 == 0, "weight chunk in whole pieces");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    volatile unsigned *full = reinterpret_cast<volatile unsigned *>(smem + NS * SLOT);   // [NS] chunks landed in the slot so far
    unsigned *freec = const_cast<unsigned *>(full) + NS;                                  // [NS] compute-wave releases of the slot
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, kh = lane >> 5;
    for (int i = tid; i < NS * SLOT; i += (NCW + NLW) * 64) smem[i] = (float)((i * 7 + blockIdx.x) % 13) * 0.01f;
    if (tid < 2 * NS) const_cast<unsigned *>(full)[tid] = 0;
    __syncthreads();
    // HBM: every chunk's activation tile is read once (unique bytes per workgroup and chunk, as in the real kernel);
    // otherwise a 96 KiB window per workgroup that stays in L2
    const float *gbb = gb + (size_t)blockIdx.x * (HBM ? 128 * BSZ : 65536);
    if (wave >= NCW) {
        // ---------------- loader wave(s)
        const int lw = NLW == 1 ? 0 : wave - NCW;
        if (LPRIO) __builtin_amdgcn_s_setprio(3);          // few instructions, all of them on the critical path of the ring
        auto issue = [&](int t) {
            float *slot = smem + (t % NS) * SLOT;
            const float *wsrc = gw + (size_t)(t & 63) * WSZ, *bsrc = gbb + (size_t)(HBM ? (t & 127) : (t & 15)) * BSZ + MISAL;   // MISAL: 4-byte-aligned activation rows
#pragma unroll
            for (int i = 0; i < PPW; ++i) {
                const int pc = lw * PPW + i;                  // wave-uniform piece index
                if (pc < PIECES) {
                    const float *src = pc * 256 < WSZ ? wsrc + pc * 256 : bsrc + (pc * 256 - WSZ);
                    dma16(src + lane * 4, slot + pc * 256);
                }
            }
        };
        if (REG) {
            // register-staged loader: global_load_dwordx4 -> VGPRs -> ds_write_b128, one chunk ahead; TWO slots suffice
            typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
            typedef float f32x4a __attribute__((ext_vector_type(4)));
            f32x4a R[PPW];
            auto rload = [&](int t) {
                const float *wsrc = gw + (size_t)(t & 63) * WSZ, *bsrc = gbb + (size_t)(HBM ? (t & 127) : (t & 15)) * BSZ + MISAL;
#pragma unroll
                for (int i = 0; i < PPW; ++i) {
                    const int pc = min(lw + NLW * i, PIECES - 1);
                    const float *src = pc * 256 < WSZ ? wsrc + pc * 256 : bsrc + (pc * 256 - WSZ);
                    R[i] = *reinterpret_cast<const f32x4u *>(src + lane * 4);
                }
            };
            auto rcommit = [&](int t) {
                float *slot = smem + (t % 2) * SLOT;
#pragma unroll
                for (int i = 0; i < PPW; ++i) {
                    const int pc = min(lw + NLW * i, PIECES - 1);
                    *reinterpret_cast<f32x4a *>(slot + pc * 256 + lane * 4) = R[i];
                }
            };
            rload(0); rcommit(0);
            if (nchunks > 1) rload(1);
            for (int h = 0; h < nchunks; ++h) {
                __syncthreads();                              // barrier h: chunk h visible (our ds_writes are drained), slot (h + 1) % 2 free
                if (h + 1 < nchunks) rcommit(h + 1);
                if (h + 2 < nchunks) rload(h + 2);
            }
            return;
        }
        if (SYNC == 0) {
            issue(0);
            if (nchunks > 1) issue(1);
            for (int h = 0; h < nchunks; ++h) {
                // chunk h has landed when at most the DMAs of chunk h + 1 are outstanding
                if (h + 1 < nchunks) {
                    if (PPW == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
                    else if (PPW == 26) asm volatile("s_waitcnt vmcnt(26)" ::: "memory");
                    else if (PPW == 52) asm volatile("s_waitcnt vmcnt(52)" ::: "memory");
                    else if (PPW == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                    else if (PPW == 13) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                 // barrier h: chunk h visible, slot of chunk h - 1 free
                if (h + 2 < nchunks) issue(h + 2);
            }
        } else {
            for (int t = 0; t < nchunks; ++t) {
                if (t >= NS) {
                    const unsigned need = (unsigned)NCW * (unsigned)(t / NS);
                    while (lds_peek(freec + t % NS) < need) __builtin_amdgcn_s_sleep(2);
                }
                issue(t);
                if (t >= 1) {
                    if (PPW == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
                    else if (PPW == 26) asm volatile("s_waitcnt vmcnt(26)" ::: "memory");
                    else if (PPW == 52) asm volatile("s_waitcnt vmcnt(52)" ::: "memory");
                    else if (PPW == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                    else if (PPW == 13) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (lane == 0) lds_bump(full + (t - 1) % NS);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) lds_bump(full + (nchunks - 1) % NS);
        }
        return;
    }
    // ---------------- compute waves
    const int offA = (wave & 1) * 64 + l31, off0 = (wave >> 1) * 64 + l31, off1 = off0 + 32;
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int g = 0; g < 16; ++g) acc[a][b][g] = 0.f;
    for (int h = 0; h < nchunks; ++h) {
        const float *Wc = smem + (h % (REG ? 2 : NS)) * SLOT, *Bc = Wc + WSZ;
        if (SYNC == 0) {
            __builtin_amdgcn_s_barrier();
        } else {
            const unsigned need = (unsigned)NLW * (unsigned)(h / NS + 1);
            while (full[h % NS] < need) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        asm volatile("" ::: "memory");
        if (PRIO) __builtin_amdgcn_s_setprio(1);
        for (int r = 0; r < TAPS; ++r) {
            const float *wr = Wc + r * (HC * MT) + offA + kh * MT, *br = Bc + r * 25 + kh * LDB;
#pragma unroll
            for (int s2 = 0; s2 < HC / 2; ++s2) {
                const float a0 = wr[2*s2*MT], a1 = wr[2*s2*MT+32], b0 = br[2*s2*LDB+off0], b1 = br[2*s2*LDB+off1];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
        }
        if (PRIO) __builtin_amdgcn_s_setprio(0);
        if (SYNC == 1) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // this wave's LDS reads of the slot are done
            if (lane == 0) atomicAdd(freec + h % NS, 1u);
        }
    }
    float s = 0.f;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int g = 0; g < 16; ++g) s += acc[a][b][g];
    out[(size_t)blockIdx.x * NCW * 64 + tid] = s;
}

__global__ void fill_kernel(float *p, size_t n, int random) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = random ? (float)(h & 0xffff) * (1.f / 65536.f) - 0.5f : 0.f;
    }
}
static int g_random = 1;     // operands: random (default; the chip holds a lower clock on random data than on zeros) or zeros (argv[1] = 0)
static void fill(float *p, size_t n) { hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, p, n, g_random); }

template <int NCW, int NLW, int HC, int LDB, int SYNC, int PRIO, int LPRIO = 0, int MISAL = 0, int HBM = 0, int REG = 0>
double run_lw(int chunks8, int blocks, size_t lds_extra = 0) {
    constexpr int WSZ = TAPS * HC * MT, BSZ = (HC * LDB + 255) / 256 * 256, SLOT = WSZ + BSZ;
    const int nchunks = chunks8 * (8 / HC);
    float *out, *gw, *gb;
    hipMalloc(&out, (size_t)blocks * NCW * 64 * 4);
    hipMalloc(&gw, (size_t)64 * WSZ * 4 + 65536); fill(gw, (size_t)64 * WSZ + 16384);
    const size_t per_wg = HBM ? 128 * BSZ : 65536;
    hipMalloc(&gb, (size_t)blocks * per_wg * 4 + (1 << 20)); fill(gb, (size_t)blocks * per_wg + (1 << 18));
    const size_t lds = (size_t)3 * SLOT * 4 + 64 + lds_extra;
    auto kern = lw_kernel<NCW, NLW, HC, LDB, SYNC, PRIO, LPRIO, MISAL, HBM, REG>;
    if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { printf("lds %zu refused\n", lds); return 0; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(blocks), dim3((NCW + NLW) * 64), lds, 0, out, gw, gb, nchunks);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(hipGetLastError())); return 0; }
    hipEventRecord(e0);
    for (int it = 0; it < 5; ++it) hipLaunchKernelGGL(kern, dim3(blocks), dim3((NCW + NLW) * 64), lds, 0, out, gw, gb, nchunks);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    hipFree(out); hipFree(gw); hipFree(gb);
    const double flops = (double)blocks * NCW * nchunks * TAPS * (HC / 2) * 4 * 4096.0;
    return flops / (ms * 1e-3) / 1e12;
}

int main(int argc, char **argv) {
    const int chunks = 64;
    if (argc > 1) g_random = atoi(argv[1]);
    printf("operands: %s\n", g_random ? "random" : "zeros");
    // shape A: 2 workgroups per CU (3 x 24 KiB slots each); activation tile from an L2-resident window (rounds 5a/5b)
    printf("A  4+1 waves, 4-ch half-chunks, 2 WG/CU, barrier, loader prio 0 / 3:   %.1f / %.1f TFLOP/s\n", run_lw<4, 1, 4, 376, 0, 0>(chunks, 512 * 6), run_lw<4, 1, 4, 376, 0, 0, 1>(chunks, 512 * 6));
    printf("A  4+2 waves, barrier, loader prio 0 / 3:                              %.1f / %.1f TFLOP/s\n", run_lw<4, 2, 4, 376, 0, 0>(chunks, 512 * 6), run_lw<4, 2, 4, 376, 0, 0, 1>(chunks, 512 * 6));
    printf("A  4+2 waves, loader prio 3, activation rows misaligned by 1 float:    %.1f TFLOP/s\n", run_lw<4, 2, 4, 376, 0, 0, 1, 1>(chunks, 512 * 6));
    // the same with every activation tile read ONCE from HBM (as the real kernel does): LDS-DMA loaders
    printf("A  HBM tiles, LDS-DMA loaders 4+1 / 4+2 / 4+4 (prio 3, misaligned):    %.1f / %.1f / %.1f TFLOP/s\n", run_lw<4, 1, 4, 376, 0, 0, 1, 1, 1>(chunks, 512 * 6),
           run_lw<4, 2, 4, 376, 0, 0, 1, 1, 1>(chunks, 512 * 6), run_lw<4, 4, 4, 376, 0, 0, 1, 1, 1>(chunks, 512 * 6));
    // register-staged loaders (global_load_dwordx4 -> ds_write_b128, one chunk ahead, two slots)
    printf("A  L2 tiles,  register loaders 4+2 / 4+4 (prio 3, misaligned):        %.1f / %.1f TFLOP/s\n", run_lw<4, 2, 4, 376, 0, 0, 1, 1, 0, 1>(chunks, 512 * 6),
           run_lw<4, 4, 4, 376, 0, 0, 1, 1, 0, 1>(chunks, 512 * 6));
    printf("A  HBM tiles, register loaders 4+1 / 4+2 / 4+4 (prio 3, misaligned):   %.1f / %.1f / %.1f TFLOP/s\n", run_lw<4, 1, 4, 376, 0, 0, 1, 1, 1, 1>(chunks, 512 * 6),
           run_lw<4, 2, 4, 376, 0, 0, 1, 1, 1, 1>(chunks, 512 * 6), run_lw<4, 4, 4, 376, 0, 0, 1, 1, 1, 1>(chunks, 512 * 6));
    printf("A  HBM tiles, register loaders 4+2 / 4+4, loader prio 0:               %.1f / %.1f TFLOP/s\n", run_lw<4, 2, 4, 376, 0, 0, 0, 1, 1, 1>(chunks, 512 * 6),
           run_lw<4, 4, 4, 376, 0, 0, 0, 1, 1, 1>(chunks, 512 * 6));
    // shape C: 1 workgroup per CU, 128 x 256 tile, 8-channel chunks (3 x 52 KiB slots)
    printf("C  8+2 / 8+4 waves, L2 tiles, LDS-DMA, barrier, loader prio 3:         %.1f / %.1f TFLOP/s\n", run_lw<8, 2, 8, 504, 0, 0, 1>(chunks, 256 * 6), run_lw<8, 4, 8, 504, 0, 0, 1>(chunks, 256 * 6));
    printf("C  8+2 / 8+4 waves, HBM tiles, LDS-DMA:                                %.1f / %.1f TFLOP/s\n", run_lw<8, 2, 8, 504, 0, 0, 1, 1, 1>(chunks, 256 * 6), run_lw<8, 4, 8, 504, 0, 0, 1, 1, 1>(chunks, 256 * 6));
    printf("C  8+2 / 8+4 waves, HBM tiles, register loaders:                       %.1f / %.1f TFLOP/s\n", run_lw<8, 2, 8, 504, 0, 0, 1, 1, 1, 1>(chunks, 256 * 6), run_lw<8, 4, 8, 504, 0, 0, 1, 1, 1, 1>(chunks, 256 * 6));
    return 0;
}
