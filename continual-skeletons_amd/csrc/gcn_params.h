// gcn_params.h -- launch parameters shared by the GCN-stage kernels (gcn.hip, gcn_dense.hip)
#pragma once
#include <stdint.h>

struct GcnParams {
    const float *x, *w, *bias;
    float *y;
    const int32_t *ell_src;
    const float *ell_val;
    int ell_cnt[3];
    int ell_w;
    int64_t adj_seg_stride, x_seg_stride, x_chan_stride, y_seg_stride, y_chan_stride;
    int Cin, CinPad, Cout, Mpad, frames, V, R, res_mode, ldb;
    unsigned vmagic, mtiles, qtiles;
    int dense;   // src[e] == e for all subsets and columns (checked on the host side of the ABI by construction)
    int adj_per_frame;   // the (dense) adjacency varies per FRAME of a segment: index = seg * frames + frame
    int lds_frames;      // frames of adjacency staged per workgroup in that mode
    int fast_epi;        // channel strides fit the 32-bit lane offsets of the scalar-base epilogue addressing
    int no_pair_reads;   // diagnostic (CSK_NO_PAIR_READS): general kernel aggregates with scalar LDS reads for even V too
    int no_vec;          // diagnostic (CSK_GCN_NOVEC): sparse kernel stages activations element-wise on every tile
    // split-K (latency mode, csk_gcn_stage_splitk_f32): split ks of a tile covers channels [ks * cper, + cper) and writes raw
    // partial sums to part[(seg * ksplit + ks)][Cout][y_chan_stride]; gcn_reduce_kernel adds them up in split order
    int ksplit, cper;
    float *part;
    // step16.hip: segment s of the launch is slot (ring_slot0 + s) % ring_slots of x / y (plain calls: no wrap, 1 << 30 slots)
    int x_ring_slots, x_ring_slot0, y_ring_slots, y_ring_slot0;
    int stagger;         // step16.hip: start delay of the odd-slot workgroup of a CU, x 64 cycles
    unsigned long long *stamps;   // step16.hip diagnostic (CSK_STAMPS under CSK_DIAG=1): s_memtime phase sums per wave, tools/stamp16_probe.py
};

// gcn_dense.hip: dense (per-segment or per-frame) adjacency with an even joint count V <= 18; returns -2 when the shape is
// not one it is built for (the caller then uses the kernels of gcn.hip), otherwise the launch status
int csk_launch_gcn_dense2(GcnParams p, int n_seg, void *stream);

// step16.hip: the slot-balanced 16x16x4 tiles for skeleton-sparse adjacencies on 16-byte-aligned layouts; -2 when the shape is
// not supported or the 32x32x2 kernels pack the chip as well (bitwise the same results either way)
int csk_launch_gcn16(GcnParams p, int n_seg, void *stream);
