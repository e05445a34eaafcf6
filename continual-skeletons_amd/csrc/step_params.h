// step_params.h -- launch parameters shared by the continual-step kernels (step.hip: 32x32x2 tiles; step16.hip: the
// slot-balanced 16x16x4 tiles)
#pragma once
#include <stdint.h>

#include "../../include/cskel.h"

struct StepParams {
    const float *ring, *w, *xres, *wres, *bias;     // xres / out are RING bases; slots are picked per emission
    float *out;
    int C, Cpad, Cout, Mpad, K, slots, head, head_step;
    int res_mode, Cres, CresPad, relu;
    int xres_slots, xres_slot0, xres_step, out_slots, out_slot0;
    int fast_epi;      // P fits the 32-bit lane byte offsets of the scalar-base epilogue addressing
    unsigned gx, gy, gz;   // position tiles, m-tiles, emission groups [* ksplit] of the launch (the grid is 1-D)
    int ksplit, cper;  // split-K (latency mode): emission groups * ksplit slices, split ks covers channels [ks*cper, ..+cper)
    float *part;       // and writes raw partial sums to part[(emission*ksplit + ks)][Cout][P]; 1 = off
    int64_t P;
    int stagger;       // step16.hip: start delay of the odd-slot workgroup of a CU, x 64 cycles
    unsigned long long *stamps;   // step16.hip diagnostic (env CSK_STAMPS=<device ptr> under CSK_DIAG=1): s_memtime stamps, tools/stamp16_probe.py
};

// step16.hip: the slot-balanced tile family (64 channels x 16*NB columns, v_mfma_f32_16x16x4_f32).  Both return -2 when the
// launch shape is not one they are built for or the policy prefers the 32x32x2 kernels (the caller then launches those);
// otherwise the launch status.  Bitwise the same results as the kernels they stand in for (same fp32 summation order).
int csk_launch_tcn_step16(StepParams p, int n_emit, void *stream);
int csk_step16_enabled();   // 0 only under CSK_DIAG=1 CSK_STEP16=1 (A/B runs)
// fused stack of 64-channel blocks (csk_co_stack_step_f32); -2: shape not supported / switched off
int csk_launch_co_stack16(int n_blocks, const csk_co_block_args *blocks, int n_skel, int V, int64_t P, void *stream);
