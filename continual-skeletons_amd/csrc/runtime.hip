// runtime.hip -- library glue of libcskel_hip.so: error text, ABI version, diagnostic switches, LDS-attribute cache.
//
// Overview of csrc/ (all compiled into libcskel_hip.so by build.sh, C ABI in include/cskel.h):
//   mfma_core.h   shared device code: fp32-MFMA "shifted GEMM" chunk, issue/commit staging (register prefetch)
//   runtime.hip   library glue: error text, ABI version, diagnostics switches, LDS-attribute cache
//   tcn.hip       tcn_stage_kernel      + csk_tcn_stage_f32        (clip TCN stage)
//   gcn.hip       gcn_stage_* kernels   + csk_gcn_stage_f32        (GCN stage: sparse fast path + general)
//   agcn.hip      agcn_attention_kernel + csk_agcn_attention_f32   (A-GCN per-sample adjacency)
//   head.hip      input norm / pooling / FC kernels and entry points
//   step.hip      continual path: tcn_step_kernel, spatial pool, window mean, logit fusion
//   executor.hip  native step executor (csk_co_plan_*)
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <unordered_map>

#include "mfma_core.h"

int csk_ensure_lds(const void *kernel, size_t bytes) {
    // the attribute is per (device, kernel); one process normally drives one GPU, but do not rely on it
    static std::mutex mu;
    static std::unordered_map<unsigned long long, size_t> cap;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    const unsigned long long key = (unsigned long long)(uintptr_t)kernel ^ ((unsigned long long)(dev + 1) << 56);
    std::lock_guard<std::mutex> lock(mu);
    auto it = cap.find(key);
    if (it != cap.end() && it->second >= bytes) return 0;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) cap[key] = bytes;
    return (int)e;
}

static thread_local char g_err[256] = "";
char *csk_err_buf() { return g_err; }
extern "C" int csk_abi_version(void) { return CSK_ABI_VERSION; }
extern "C" const char *csk_last_error(void) { return g_err; }

// Diagnostics are read from the environment per call only when CSK_DIAG is set at library load (so that the normal
// launch path never touches the environment): CSK_STAMPS=<device ptr> enables the s_memtime stamps of
// tcn_stage_kernel (tools/stamp_probe.py), CSK_GCN_GENERAL=1 forces the general (dense-capable) GCN kernel,
// CSK_NOPRIO=1 drops the raised wave priority inside MFMA segments (tools/ab_probe.py).
static const bool g_diag = getenv("CSK_DIAG") != nullptr;
bool csk_diag_flag(const char *name) { return g_diag && getenv(name) != nullptr; }
int csk_diag_int(const char *name) {
    const char *v = g_diag ? getenv(name) : nullptr;
    return v ? atoi(v) : 0;
}
unsigned long long *csk_diag_stamps() {
    if (!g_diag) return nullptr;
    const char *d = getenv("CSK_STAMPS");
    return d ? (unsigned long long *)strtoull(d, nullptr, 0) : nullptr;
}

// ---- stream concurrency probe (include/cskel.h: csk_stream_overlap_probe) -----------------------------------------
__global__ void spin_kernel(long long ticks) {
    const long long t0 = wall_clock64();                 // constant 100 MHz counter
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

extern "C" int csk_stream_overlap_probe(void *stream_a, void *stream_b, int spin_us, float *ratio) {
    if (!ratio) CSK_FAIL("stream_overlap_probe: null pointer");
    if (spin_us < 10 || spin_us > 100000) CSK_FAIL("stream_overlap_probe: spin_us must be in [10, 100000], got %d", spin_us);
    hipStream_t a = (hipStream_t)stream_a, b = (hipStream_t)stream_b;
    const long long ticks = 100LL * spin_us;
    hipEvent_t e0 = nullptr, ea = nullptr, eb = nullptr;
    hipError_t err = hipSuccess;
    float alone = 0.f, ta = 0.f, tb = 0.f;
#define CSK_TRY(x) do { err = (x); if (err != hipSuccess) goto done; } while (0)
    CSK_TRY(hipEventCreate(&e0)); CSK_TRY(hipEventCreate(&ea)); CSK_TRY(hipEventCreate(&eb));
    for (int pass = 0; pass < 2; ++pass) {               // pass 0 also absorbs the first-launch cost
        CSK_TRY(hipEventRecord(e0, a));
        spin_kernel<<<1, 64, 0, a>>>(ticks);
        CSK_TRY(hipEventRecord(ea, a));
        CSK_TRY(hipStreamSynchronize(a));
        CSK_TRY(hipEventElapsedTime(&alone, e0, ea));
    }
    CSK_TRY(hipStreamSynchronize(b));
    CSK_TRY(hipEventRecord(e0, a));
    CSK_TRY(hipStreamWaitEvent(b, e0, 0));
    spin_kernel<<<1, 64, 0, a>>>(ticks);
    spin_kernel<<<1, 64, 0, b>>>(ticks);
    CSK_TRY(hipEventRecord(ea, a));
    CSK_TRY(hipEventRecord(eb, b));
    CSK_TRY(hipStreamSynchronize(a));
    CSK_TRY(hipStreamSynchronize(b));
    CSK_TRY(hipEventElapsedTime(&ta, e0, ea));
    CSK_TRY(hipEventElapsedTime(&tb, e0, eb));
    *ratio = (ta > tb ? ta : tb) / (alone > 1e-6f ? alone : 1e-6f);
done:
#undef CSK_TRY
    if (e0) (void)hipEventDestroy(e0);
    if (ea) (void)hipEventDestroy(ea);
    if (eb) (void)hipEventDestroy(eb);
    return (int)err;
}
