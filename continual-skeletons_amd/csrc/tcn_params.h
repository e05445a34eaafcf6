// tcn_params.h -- launch parameters of the clip temporal-conv stage (tcn.hip: 32x32x2 tiles; tcn16.hip: 16x16x4 tiles)
#pragma once
#include <stdint.h>

struct TcnParams {
    const float *y, *w, *xres, *wres, *bias;
    float *out;
    int C, Cpad, Cout, Mpad, Tin, Tout, V, K, stride, pad;
    int res_mode, Cres, CresPad, Tres, res_off, relu, ldb;
    unsigned vmagic, mtiles, qtiles;
    int nt;                       // positions per tile actually used (<= 16384 / MT)
    int fast_epi;                 // row strides fit the 32-bit lane offsets of the scalar-base epilogue addressing
    int prio;                     // raise wave priority inside MFMA segments (diagnostic CSK_NOPRIO=1 turns it off)
    int vec_stage;                // 16-byte activation staging on interior tiles (diagnostic CSK_TCN_NOVEC=1 turns it off)
    int no_peel_ct;
    int ldb2;                     // conv-residual phase: LDS row stride of its activation tile
    int ksplit, cper;             // split-K form (csk_tcn_stage_splitk_f32): ksplit channel ranges of cper channels per tile,
    float *part;                  // raw partial sums part[(seg * ksplit + ks)][Cout][Tout * V]; ksplit == 1: off
    unsigned long long *stamps;   // diagnostic (env CSK_STAMPS=<device ptr>): s_memtime stamps per workgroup, see tools/stamp_probe.py
};

// tcn16.hip: the 16x16x4 tile family for the 9-tap temporal conv at V = 25 / 18 and stride 1 / 2; -2 when the shape is not one
// it is built for (the caller launches the 32x32x2 kernels).  Bitwise the same sums (same (chunk, tap, channel) order).
int csk_launch_tcn_stage16(TcnParams p, int n_seg, void *stream);
