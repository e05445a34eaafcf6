/*
 * cskel.h -- C ABI of the MI355X-native ST-GCN / CoST-GCN forward path (libcskel_hip.so).
 *
 * The reference (LukasHedegaard/continual-skeletons) has no native code and no FFI: its hot path is
 * the Python nn.Module surface of models/base.py issuing stock ATen ops.  Each entry point below
 * replaces the ATen op sequence of one reference method (cited per function); the Python host layer
 * in continual-skeletons_amd/ mirrors the reference's module interface and binds these with ctypes.
 *
 * Conventions
 *  - all tensors fp32, device (HBM) pointers, laid out exactly as the reference's contiguous tensors:
 *    clip activations (NM, C, T, V) with V innermost; "segment" = one of the NM skeleton sequences.
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream); launches are asynchronous.  The stage / step /
 *    head entry points (csk_gcn_stage_f32 ... csk_fuse_rank_f32) neither allocate nor synchronise and are
 *    graph-capture safe (the first launch of a kernel instantiation raises its LDS cap with hipFuncSetAttribute).
 *    Exceptions: csk_stream_overlap_probe synchronises both streams it is given; csk_co_plan_create / _destroy
 *    allocate host memory.
 *  - packed weights are produced once on the host by continual-skeletons_amd/fold.py
 *    (BatchNorm(eval) + bias folding, zero padding of channel counts to CSK_CPAD / CSK_MT multiples).
 *  - return value: 0 = ok; <0 = argument error (see csk_last_error()); >0 = hipError_t of the launch.
 */
#ifndef CSKEL_H
#define CSKEL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CSK_ABI_VERSION 15
#define CSK_KC 8    /* channel-chunk of the K loop of the TCN kernels                                */
#define CSK_CPAD 16 /* packed weights zero-pad C_in to a multiple of this                             */
#define CSK_MT 64  /* packed weights pad C_out to a multiple of this                               */

/* residual forms of SpatioTemporalBlock (models/base.py:367-374) and GraphConvolution (:246-254) */
#define CSK_RES_NONE 0
#define CSK_RES_IDENTITY 1
#define CSK_RES_CONV 2

int csk_abi_version(void);
/* thread-local text of the last argument error */
const char *csk_last_error(void);

/*
 * Do two HIP streams execute concurrently on this device?  HIP multiplexes its streams onto a few hardware queues
 * (4 by default); two streams that land on the same queue run strictly one after the other, which silently turns the
 * stream shards of the continual path (parallel.py:StreamShards; the reference has no counterpart, it steps one
 * Python module at a time) into a serial schedule.  Launches one single-wavefront kernel that spins for spin_us
 * microseconds on each stream behind a common start event and writes
 *   *ratio = (time until both have finished) / (time of one alone):   ~1 = concurrent, ~2 = serialised.
 * Synchronises both streams.  No other work should be in flight on either stream.
 */
int csk_stream_overlap_probe(void *stream_a, void *stream_b, int spin_us, float *ratio);

/*
 * GraphConvolution.forward, models/base.py:260-270 (and its per-frame use by CoGraphConvolution,
 * base.py:273-276):   y = ReLU( BN( sum_i W_i * (x .A_eff[i]) + b_i ) + gcn_residual(x) )
 *
 *  x        (n_seg, c_in, frames, V) with strides x_seg_stride / x_chan_stride (elements); frames
 *           contiguous (V floats each).
 *  y        (n_seg, c_out, frames, V), same convention.
 *  w        packed [R][c_in_pad][c_out_pad]; R = 3 subsets (+1 = conv gcn_residual when
 *           res_mode == CSK_RES_CONV); BN scale folded in.
 *  bias     [c_out_pad] folded (conv biases, BN shift, and gcn_residual's BN shift).
 *  ell_src  [3][V][ell_w] int32 : row indices v of the non-zeros of column w of A_eff[i] (padded with 0)
 *  ell_val  [3][V][ell_w] fp32  : their values (padded with 0.0)
 *  ell_cnt  [3] : number of meaningful entries per subset (<= ell_w) -- lets sparse subsets skip padding
 *  adj_seg_stride: 0 for a graph shared by all segments (ST-GCN); 3*V*ell_w for per-segment VALUES
 *           (A-GCN's per-sample attention, models/a_gcn/a_gcn.py:62-65; ell_src stays shared).  A per-segment
 *           adjacency with ell_w == V and ell_cnt == {V,V,V} MUST list every source joint in order
 *           (ell_src[i][w][e] == e): the kernel then takes its dense fast path and does not read ell_src.
 *  adj_per_frame: 0, or 1 = the dense adjacency varies per FRAME: matrix index = seg*frames + frame, consecutive
 *           matrices adj_seg_stride apart (continual A-GCN: every frame of the channel-major layout is another
 *           skeleton with its own attention, models/coa_gcn/coa_gcn.py:11-14).
 *  res_mode CSK_RES_IDENTITY (c_in == c_out) or CSK_RES_CONV.
 */
int csk_gcn_stage_f32(const float *x, float *y, const float *w, const float *bias,
                      const int32_t *ell_src, const float *ell_val, const int32_t *ell_cnt,
                      int ell_w, int64_t adj_seg_stride, int adj_per_frame,
                      int n_seg, int c_in, int c_out, int frames, int V,
                      int64_t x_seg_stride, int64_t x_chan_stride,
                      int64_t y_seg_stride, int64_t y_chan_stride,
                      int res_mode, void *stream);

/*
 * The same stage with its K loop split over workgroups (latency mode: a handful of streams, where one workgroup per tile
 * would walk all 3 * c_in / 8 K-chunks alone -- 41 us at c_in = 256): the channel axis is cut into up to ksplit ranges
 * computed by separate workgroups into `partial` ([n_seg * ksplit][c_out][y_chan_stride] floats) and summed in split
 * order by a second kernel that also applies bias / identity gcn_residual / ReLU.  Graphs shared by all segments with
 * <= 1 / 1 / 4 non-zeros per column only (skeleton graphs).  Results differ from csk_gcn_stage_f32 by fp32 summation
 * order only; for a given ksplit they do not depend on n_seg or frames.  ksplit = 1 is csk_gcn_stage_f32.
 */
int csk_gcn_stage_splitk_f32(const float *x, float *y, const float *w, const float *bias, const int32_t *ell_src,
                             const float *ell_val, const int32_t *ell_cnt, int ell_w, int n_seg, int c_in, int c_out,
                             int frames, int V, int64_t x_seg_stride, int64_t x_chan_stride, int64_t y_seg_stride,
                             int64_t y_chan_stride, int res_mode, int ksplit, float *partial, void *stream);

/*
 * TemporalConvolution.forward (models/base.py:302-304) fused with the tail of
 * SpatioTemporalBlock.forward (base.py:376-387):
 *     out = ReLU( BN(conv_{k x 1, stride s, pad p}(y)) + residual(x[:, :, shrink:T-shrink]) )
 * or, with relu == 0 and res_mode == NONE, a bare TemporalConvolution.
 *
 *  y        (n_seg, c, t_in, V) contiguous           -- output of the GCN stage
 *  w        packed [k][c_pad][c_out_pad], BN scale folded
 *  x_res    (n_seg, c_res, t_res, V) contiguous or NULL -- block input for the residual
 *  w_res    packed [1][c_res_pad][c_out_pad] (CSK_RES_CONV) or NULL
 *  bias     [c_out_pad] folded (t_conv bias, BN shift, residual conv bias + BN shift)
 *  out      (n_seg, c_out, t_out, V), t_out = (t_in + 2p - k)/s + 1
 *  res_off  frame offset of the residual: out[t'] pairs with x_res[t'*s + res_off]
 *           (= residual_shrink, base.py:379-383; 0 when padded)
 */
int csk_tcn_stage_f32(const float *y, const float *w, const float *x_res, const float *w_res,
                      const float *bias, float *out,
                      int n_seg, int c, int c_out, int t_in, int V, int k, int stride, int pad,
                      int res_mode, int c_res, int t_res, int res_off, int relu, void *stream);

/*
 * A whole SpatioTemporalBlock with a FEW input channels and no block residual -- layer 1 of the reference's stacks
 * (models/st_gcn/st_gcn.py:30: StGcnBlock(3, 64, A, residual=False); block body models/base.py:376-387) -- in ONE launch:
 *   out = ReLU( tcn( gcn(x) ) ),  gcn(x) = ReLU( sum_k W'_k . (x . A_k) + b' + conv1x1+BN(x) )   (models/base.py:230-270)
 * csk_gcn_stage_f32 (conv gcn_residual) followed by csk_tcn_stage_f32 (k = 9, stride 1, pad 4, no residual), with the graph
 * conv formed on the fly inside the temporal conv's tile: y never travels through memory (2 GB per batch-256 forward).  The
 * results are BIT FOR BIT those of the two calls (the same fmaf chain in the graph conv's K order; zero in the frames of the
 * temporal padding).
 *  x        (n_seg, c_in, t_in, V) contiguous, 1 <= c_in <= 4;  out (n_seg, c_out, t_in, V)
 *  gcn_w / gcn_bias / ell_*  the packed operands of csk_gcn_stage_f32 with the conv gcn_residual as 4th subset, c_mid outputs
 *  tcn_w / tcn_bias          the packed operands of csk_tcn_stage_f32 (c_mid -> c_out, 9 taps)
 *  V in {25, 18}; c_mid a multiple of 8; a skeleton-sparse adjacency (ell_cnt <= 1 / 1 / 4).  Other shapes: the two calls.
 */
int csk_block_few_channels_f32(const float *x, const float *gcn_w, const float *gcn_bias, const int32_t *ell_src,
                               const float *ell_val, const int32_t *ell_cnt, int ell_w, const float *tcn_w,
                               const float *tcn_bias, float *out, int n_seg, int c_in, int c_mid, int c_out, int t_in,
                               int V, int pad, void *stream);

/*
 * csk_tcn_stage_f32 for launches of a FEW tiles (small-batch clip inference; the reference's own CPU protocol is batch 1,
 * scripts/benchmark_all_ntu60.py:17): the K loop of every output tile is cut into `ksplit` channel ranges walked by
 * separate workgroups; raw partial sums go to `partial` ([n_seg * ksplit][c_out][t_out * V] floats, 16-byte aligned) and a
 * second launch adds them IN SPLIT ORDER, then bias, identity residual and ReLU (the conv residual rides in split 0).
 * Same method and arguments as csk_tcn_stage_f32 (models/base.py:302-304 + 376-387); k must be 9.  For a given ksplit the
 * result of a clip does not depend on n_seg; against ksplit = 1 it differs by the summation order only.
 */
int csk_tcn_stage_splitk_f32(const float *y, const float *w, const float *x_res, const float *w_res,
                             const float *bias, float *out,
                             int n_seg, int c, int c_out, int t_in, int V, int k, int stride, int pad,
                             int res_mode, int c_res, int t_res, int res_off, int relu, int ksplit, float *partial,
                             void *stream);

/*
 * Bare 1 x 1 conv + bias on the layouts of csk_gcn_stage_f32 (no adjacency, no residual, no ReLU): the six a_i / b_i
 * embedding convs of AdaptiveGraphConvolution fused into one GEMM, models/a_gcn/a_gcn.py:27-28, 53-59.
 *  x (n_seg, c_in, frames, V), y (n_seg, c_out, frames, V) with explicit segment / channel strides (elements);
 *  w packed [1][c_in_pad][c_out_pad], bias [c_out_pad].
 */
int csk_conv1x1_f32(const float *x, float *y, const float *w, const float *bias, int n_seg, int c_in, int c_out,
                    int frames, int V, int64_t x_seg_stride, int64_t x_chan_stride, int64_t y_seg_stride,
                    int64_t y_chan_stride, void *stream);

/*
 * OPT-IN precision mode "bf16x3" of csk_tcn_stage_f32 (same reference method, models/base.py:302-304 + 376-387; same
 * arguments and layouts except the weights): the 9 x 1 temporal conv and the 1 x 1 residual conv run on the bf16 matrix
 * pipe with every fp32 operand split into three bf16 pieces (h + m + l, 24 significand bits) and the six piece products
 * of order <= 2 accumulated in fp32 -- fp32-GRADE results (measured max error vs the oracle: profiles/r03_parity_report.json),
 * not the exact fp32 arithmetic of csk_tcn_stage_f32.  Never selected implicitly: blocks.set_precision(module, "bf16x3").
 *  w_split      packed split weights of the (k = 9) conv, BN scale folded: 16-byte vectors of 8 bf16 (8 consecutive
 *               input channels) indexed [c_pad / 16][9 tap slots][3 pieces][2 channel halves][c_out_pad], the taps in
 *               CLASS-MAJOR order (0, s, 2s, ..., 1, 1 + s, ... for stride s: the taps of a residue class read one
 *               de-interleaved set of source frames); fold.pack_conv_weight_split
 *  w_res_split  the same for the 1 x 1 residual conv: [c_res_pad / 16][3 slots: tap 0, zero, zero][3][2][c_out_pad], or NULL
 *  k            must be 9; stride <= 4
 */
int csk_tcn_stage_bf16x3(const float *y, const void *w_split, const float *x_res, const void *w_res_split,
                         const float *bias, float *out,
                         int n_seg, int c, int c_out, int t_in, int V, int k, int stride, int pad,
                         int res_mode, int c_res, int t_res, int res_off, int relu, void *stream);

/*
 * OPT-IN precision mode "bf16x3" of csk_gcn_stage_f32 (same reference method, models/base.py:260-270) for skeleton-sparse
 * graphs (<= 1/1/4 non-zeros per adjacency column), C_out a multiple of 128 and contiguous (n_seg, C, frames, V) tensors:
 * the adjacency aggregation stays exact fp32, the channel-mixing GEMM runs on the bf16 matrix pipe with its operands split
 * into three bf16 pieces (fp32-grade, see csk_tcn_stage_bf16x3).
 *  w_split      split image of the three 1 x 1 convs (BN scale folded): [c_in_pad / 16][3 subsets][3 pieces][2 halves][c_out_pad]
 *  w_res_split  split image of the conv gcn_residual, [c_in_pad / 16][3 slots: the conv, zero, zero][3][2][c_out_pad], or NULL
 *  bias, ell_*  as csk_gcn_stage_f32 (shared adjacency: adj_seg_stride = 0)
 */
int csk_gcn_stage_bf16x3(const float *x, float *y, const void *w_split, const void *w_res_split, const float *bias,
                         const int32_t *ell_src, const float *ell_val, const int32_t *ell_cnt, int ell_w,
                         int n_seg, int c_in, int c_out, int frames, int V, int res_mode, void *stream);

/*
 * Input permute + data_bn + reshape, models/st_gcn/st_gcn.py:49-57 (clip) and
 * models/base.py:73-82 (per frame, t = 1):
 *     h[n*M+m, c, t, v] = x[n, c, t, v, m] * scale[(m*V+v)*C+c] + shift[(m*V+v)*C+c]
 *  x (N, C, T, V, M) contiguous;  h (N*M, C, T, V) with strides h_seg_stride/h_chan_stride.
 */
int csk_input_norm_f32(const float *x, const float *scale, const float *shift, float *h,
                       int N, int C, int T, int V, int M,
                       int64_t h_seg_stride, int64_t h_chan_stride, void *stream);

/*
 * Head, models/st_gcn/st_gcn.py:60-64: feat[n, c] = mean_m mean_{t,v} h[n*M+m, c, t, v];
 * logits = feat @ fc_w^T + fc_b.   h (N*M, C, TV) contiguous; feat (N, C) is also returned
 * (it is the per-step feature of CoModelBase.spatial_pool, models/base.py:84).
 *  fc_w (classes, C) row-major as in nn.Linear; logits (N, classes).  Pass logits == NULL to only pool.
 */
int csk_pool_fc_f32(const float *h, const float *fc_w, const float *fc_b, float *feat, float *logits,
                    int N, int M, int C, int TV, int classes, void *stream);

/* feat[n, c] = scale * mean_m mean_{tv} h[n*M+m, c, tv]: the clip-mode evaluation of CoModelBase's head
 * (spatial_pool + zero-padded co.AvgPool1d window, models/base.py:84-97,166-181) uses scale = frames/pool_size */
int csk_pool_scaled_f32(const float *h, float *feat, int N, int M, int C, int TV, float scale, void *stream);

/* plain FC on pooled features: logits[n] = feat[n] @ fc_w^T + fc_b  (co.Linear, models/base.py:99).  When C is a
 * multiple of 4 the rows are read 16 bytes at a time: feat and fc_w must then be 16-byte aligned (an error otherwise). */
int csk_fc_f32(const float *feat, const float *fc_w, const float *fc_b, float *logits,
               int N, int C, int classes, void *stream);

/*
 * A-GCN adaptive adjacency, models/a_gcn/a_gcn.py:53-63.  E holds the a_conv / b_conv embeddings of the three
 * subsets (channels [i*inter+k] = a_conv_i, [(3+i)*inter+k] = b_conv_i, biases included; produced by
 * csk_conv1x1_f32); element (n, ch, t, v) at
 * (n / seg_per_group)*e_group_stride + (n % seg_per_group)*e_seg_stride + ch*e_chan_stride + t*V + v
 * (clip: seg_per_group = n_seg, e_group_stride = 0; continual: a group = one channel-major frame (6*inter, P) of
 * the frames of a launch cycle, a segment = one skeleton of it, e_seg_stride = V, T = 1 -- per-frame attention,
 * models/coa_gcn/coa_gcn.py:11-14).  For every sample and subset:
 *     adj[i][v, w] = softmax_v( sum_{k,t} Ea[k,t,v] * Eb[k,t,w] / (inter*T) ) + a_sum[i][v, w],   a_sum = A + graph_attn
 * written as the dense column-wise ELL values ell_val[n][i][w][v] that csk_gcn_stage_f32 consumes with
 * ell_w = V, ell_cnt = {V,V,V}, adj_seg_stride = 3*V*V.
 * scratch: n_seg * 3 * 4 * V * V floats of partial logits (clip form, T > 1: the K = inter*T contraction is cut into 4
 * channel ranges per sample, summed in a fixed order); may be NULL for T == 1.  In the clip form E must be readable
 * 12 bytes past its last element (16-byte loads of the last partial row vector) and 4-byte aligned rows suffice.
 */
int csk_agcn_attention_f32(const float *E, const float *a_sum, float *ell_val, float *scratch, int n_seg, int inter, int T,
                           int V, int64_t e_seg_stride, int64_t e_chan_stride, int seg_per_group, int64_t e_group_stride,
                           void *stream);

/* The two entries above fused (no E tensor in memory): the six embedding 1x1 convs and the attention.
 * x: n_seg segments of `frames` frames (element (seg, c, frame, v) at seg*x_seg_stride + c*x_chan_stride + frame*V + v); w_pairs / b_pairs: the embedding weights packed as for csk_conv1x1_f32 ([C_in_pad][C_out_pad], C_out =
 * 6*inter) but in PAIR-MAJOR row order: rows i*2*inter + k = a_conv_i[k], i*2*inter + inter + k = b_conv_i[k].
 * per_frame = 1 (CoAGCN step form, models/coa_gcn/coa_gcn.py:11-14: the module applied per frame, T = 1; a segment is a
 *   channel-major slot, its "frames" are skeletons): one attention per (segment, frame), ell_val[(seg*frames + frame)][i][w][v].
 * per_frame = 0 (A-GCN clip form, models/a_gcn/a_gcn.py:53-63): one attention per segment over all frames, ell_val[seg][i][w][v];
 *   scratch: n_seg * 3 * ceil(frames / (128 / V)) * V * V floats (partial logits per tile of 128 / V frames, summed in tile order).
 * Built for V in {18, 25} and inter in {16, 32, 64}; fails otherwise (callers then use the two entries above).  Even V: rows
 * 8-byte aligned. */
int csk_agcn_embed_attention_f32(const float *x, const float *w_pairs, const float *b_pairs, const float *a_sum, float *ell_val,
                                 float *scratch, int n_seg, int c_in, int inter, int frames, int V, int per_frame,
                                 int64_t x_seg_stride, int64_t x_chan_stride, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Continual (frame-by-frame) path.  The arithmetic the reference delegates to the third-party package
 * continual-inference (co.Conv2d / co.Delay / co.Residual / co.AvgPool1d, call sites models/base.py:73-101,
 * 273-276, 307-334, 390-446) on per-module Python-side buffers runs here on a persistent HBM state slab in
 * CHANNEL-MAJOR layout: a frame of activations for all streams is a (C, P) matrix, P = streams*M*V positions
 * rounded up to a multiple of 4, joint innermost.  csk_gcn_stage_f32 handles the per-frame graph conv on that
 * layout (n_seg = 1, frames = streams*M, x_chan_stride = P).
 * ------------------------------------------------------------------------------------------------ */

/*
 * Emitting step(s) of CoTemporalConvolution (+ residual + ReLU of CoSpatioTemporalBlock, base.py:412-446).
 * For emission j = 0 .. n_emit-1 (n_emit > 1 batches the frames of a stride cycle; the kernel folds 2 or 4 emissions
 * into one workgroup tile when their count allows, so that one staged ring window serves all their taps):
 *     h_j   = (head + j*head_step) mod slots                       slot of the newest post-GCN frame
 *     out_j[co, p] = ReLU( sum_r sum_c W[r][c][co] * ring[(h_j-(k-1)+r) mod slots][c][p] + bias[co] + res_j[co, p] )
 *  ring   [slots][c][P] post-GCN frames; zero-initialised slots act as the clip conv's zero padding;
 *         slots >= k + (n_emit-1)*head_step.
 *  x_res  ring [x_res_slots][c_res][P] of block inputs or NULL; emission j pairs with slot
 *         (x_res_slot0 + j*x_res_step) mod x_res_slots = the input delayed by (k-1)/2 frames (co.Delay).
 *  out    ring [out_slots][c_out][P]; emission j is written to slot (out_slot0 + j) mod out_slots.
 *  All ring bases 16-byte aligned, P % 4 == 0.
 *  ksplit > 1: the channel axis is cut into up to ksplit ranges computed by separate workgroups into `partial`
 *  ([n_emit * ksplit][c_out][P] floats, 16-byte aligned) and summed in a fixed order by a second kernel that also
 *  applies bias / identity residual / ReLU.  Used in latency mode for few streams, where one workgroup per tile would walk
 *  the whole K loop alone (rounds 2-5 also split the 256-channel blocks in 3 in the default mode; the slot-balanced tiles of
 *  csrc/step16.hip made that unnecessary).  Results differ from ksplit = 1 by fp32
 *  summation order only; for a given ksplit they do not depend on n_emit or P.
 */
int csk_tcn_step_f32(const float *ring, int slots, int head, int head_step, int n_emit, const float *w,
                     const float *x_res, int x_res_slots, int x_res_slot0, int x_res_step,
                     const float *w_res, const float *bias, float *out, int out_slots, int out_slot0,
                     int c, int c_out, int64_t P, int k, int res_mode, int c_res, int relu, int ksplit, float *partial,
                     void *stream);

/*
 * One CoSpatioTemporalBlock.forward_step cycle in ONE launch (models/base.py:412-446): graph conv of the 4 new frames of
 * a stride cycle + the 4 emitting steps of the temporal conv + residual + ReLU -- csk_gcn_stage_f32 followed by
 * csk_tcn_step_f32 with n_emit = 4, fused (same arithmetic, same summation order, bit-identical results).  For blocks
 * with c_out <= 64, temporal stride 1, k = 9, a skeleton-sparse adjacency (ell_cnt <= 1/1/4) and a block residual that
 * is absent (CSK_RES_NONE) or the identity; all 4 frames must emit (the block has seen >= 4 frames before).
 *  xin     block input ring [xin_slots][c_in][P]; new frame f = 0..3 in slot (xin_slot0 + f) mod xin_slots; the residual
 *          frame of emission j in slot (x_res_slot0 + j) mod xin_slots (the input delayed by 4 frames)
 *  y_ring  [y_slots][c_out][P] post-GCN frames, y_slots >= 12; new frame f is WRITTEN to slot (y_slot0 + f) mod y_slots,
 *          the temporal window of emission j is slots y_slot0 + j - 8 .. y_slot0 + j
 *  out     [out_slots][c_out][P]; emission j goes to slot (out_slot0 + j) mod out_slots
 *  n_skel  skeletons per frame (streams * M); P >= n_skel * V, a multiple of 4
 *  gcn_* / ell_* / tcn_*: the packed operands of csk_gcn_stage_f32 / csk_tcn_step_f32.
 */
int csk_co_block_step_f32(const float *xin, int xin_slots, int xin_slot0, int c_in, const float *gcn_w,
                          const float *gcn_bias, const int32_t *ell_src, const float *ell_val, const int32_t *ell_cnt,
                          int ell_w, int gcn_res_mode, float *y_ring, int y_slots, int y_slot0, const float *tcn_w,
                          const float *tcn_bias, int res_mode, int x_res_slot0, float *out, int out_slots, int out_slot0,
                          int c_out, int n_skel, int V, int64_t P, void *stream);

/*
 * A STACK of consecutive blocks of that kind in one launch (round 6): csk_co_block_step_f32 for block 0, then for block 1 on
 * block 0's output ring, ... -- n_blocks <= CSK_CO_STACK_MAX.  Block i + 1's `xin` must be block i's `out` (same slot counts),
 * its new frames the slots block i emits into.  A workgroup owns the same positions (whole skeletons) of the four frames
 * through every block: no workgroup depends on another (step mode has no temporal halo), stage outputs travel through the
 * state rings and L2.  Bitwise the results of the per-block calls.  Needs the slot-balanced tile family (csrc/step16.hip:
 * V = 25 or 18 joints, c_out a multiple of 4, rings below 4 GB); otherwise (or with one block) it IS the per-block calls.
 */
#define CSK_CO_STACK_MAX 4
typedef struct csk_co_block_args {          /* the arguments of csk_co_block_step_f32 that differ per block */
    const float *xin;
    int32_t xin_slots, xin_slot0, c_in;
    const float *gcn_w, *gcn_bias;
    const int32_t *ell_src;
    const float *ell_val;
    int32_t ell_cnt[3], ell_w, gcn_res_mode;
    float *y_ring;
    int32_t y_slots, y_slot0;
    const float *tcn_w, *tcn_bias;
    int32_t res_mode, x_res_slot0;
    float *out;
    int32_t out_slots, out_slot0, c_out;
} csk_co_block_args;
int csk_co_stack_step_f32(int n_blocks, const csk_co_block_args *blocks, int n_skel, int V, int64_t P, void *stream);

/* spatial_pool of CoModelBase (models/base.py:84) on a channel-major frame: feat[n, c] = mean of the MV = M*V
 * positions of stream n.  h (C, P); feat (N, C). */
int csk_co_spatial_pool_f32(const float *h, float *feat, int N, int C, int MV, int64_t P, void *stream);

/* co.AvgPool1d(window, stride 1) step (models/base.py:97): pooled = (1/window) * sum of the `count` newest
 * entries of ring [window][n_elem] (newest at slot `head`); missing entries count as zeros. */
int csk_co_window_mean_f32(const float *ring, float *pooled, int64_t n_elem, int window, int head, int count,
                           void *stream);

/* The three steps above and csk_fc_f32 in ONE launch (one workgroup per stream), bit-identical to issuing them one by one:
 * spatial_pool of frame h (NULL: a zero feature, the window's end padding) into pool_ring[head], then -- when `emit` --
 * pooled = window mean of the `count` newest entries and logits = pooled @ fc_w^T + fc_b.  pool_ring [window][N][C],
 * pooled [N][C], logits [N][classes]. */
int csk_co_head_step_f32(const float *h, float *pool_ring, float *pooled, const float *fc_w, const float *fc_b, float *logits,
                         int N, int C, int MV, int64_t P, int window, int head, int count, int emit, int classes, void *stream);

/* csk_input_norm_f32 for the r = 1..8 frames of a launch cycle in one launch (continual form, T = 1): frames[f] (N, C, V, M)
 * -> dst[f] channel-major (C, P) (models/base.py:73-82).  frames / dst are HOST arrays of device pointers. */
int csk_input_norm_frames_f32(const float *const *frames, float *const *dst, int r, const float *scale, const float *shift,
                              int N, int C, int V, int M, int64_t P, void *stream);

/*
 * Multi-stream logit fusion + top-k support, scripts/multi_stream_eval.py:33-60: fused = left fold of add
 * (use_max = 0) or maximum (1) over n_streams <= 4 prediction arrays (host array of device pointers); element
 * (n, c) of every array at n*sample_stride + c*class_stride (a (N, classes, steps) array with the reference's
 * `preds[:, :, 0]` selection has sample_stride = classes*steps, class_stride = steps).  Outputs (either may be
 * NULL): fused (N, classes) contiguous; rank[n] = number of classes scoring strictly higher than targets[n]
 * (top-k hit <=> rank < k; out-of-range targets give rank = classes).
 */
int csk_fuse_rank_f32(const float *const *preds, int n_streams, int use_max, int N, int classes,
                      int64_t sample_stride, int64_t class_stride, const int64_t *targets, float *fused, int *rank,
                      void *stream);

/* ------------------------------------------------------------------------------------------------
 * Native step executor ("plan"): the counterpart of co.Sequential.forward_step driving the ten continual
 * blocks and the head (models/base.py:108-122,183-190) -- one C call issues every launch of a cycle of 1..8
 * frames (input norm, per block one GCN-stage + one multi-emission TCN-step launch, spatial pool, temporal
 * window mean, FC), with the ring-slot / stride-phase bookkeeping kept in the plan.  No allocation, no sync.
 * ------------------------------------------------------------------------------------------------ */
#define CSK_CO_MAX_CYCLE 8
/* Ring depths are per layer (csk_co_layer.y_slots / .out_slots, csk_co_plan_create xin0_slots), derived from the frames ONE
 * launch of the layer can receive (max_in = CSK_CO_MAX_CYCLE / cumulative stride of the layers in front of it):
 *   post-GCN ring   : the (k-1) = 8 window frames of co.Conv2d + the new frames      -> CSK_CO_Y_SLOTS(max_in)
 *   input history   : residual lag (k-1)/2 = 4 (co.Delay / residual_shrink) + the new frames -> CSK_CO_IN_SLOTS(max_in);
 *                     a layer's output ring IS the next layer's input history, so out_slots(l) = CSK_CO_IN_SLOTS(max_in(l+1))
 *                     (last layer: at least its own emissions per launch; csk_co_block_step_f32 wants >= 4).
 * A frame s lives in slot s % depth of its ring.  Deeper rings are accepted. */
#define CSK_CO_Y_SLOTS(max_in) (8 + (max_in))
#define CSK_CO_IN_SLOTS(max_in) (4 + (max_in))

typedef struct csk_co_layer {
    int32_t c_in, c_out, stride, res_kind;   /* res_kind: CSK_RES_NONE / IDENTITY / CONV (block residual) */
    int32_t gcn_res_mode, ell_w, ell_cnt[3];
    int32_t tcn_ksplit;                      /* split-K of the TCN step (csk_tcn_step_f32); 1 = off               */
    int32_t y_slots, out_slots;              /* depths of y_ring / out_ring (see CSK_CO_Y_SLOTS / CSK_CO_IN_SLOTS)  */
    int32_t partial_emits;                   /* emissions the split-K scratch holds (0 when tcn_ksplit == 1)       */
    int32_t agcn_adj_frames;                 /* frames of per-skeleton adjacencies agcn_adj holds (0: no adaptive graph conv) */
    int32_t gcn_ksplit, gcn_partial_frames;  /* split-K of the graph conv (csk_gcn_stage_splitk_f32, latency mode); <= 1 = off.  Its
                                              * partial sums go to tcn_partial as well (launches are stream-ordered), which then
                                              * holds gcn_partial_frames * gcn_ksplit * c_out * P floats: the frames ONE graph-conv
                                              * launch may cover (more frames in a cycle fail with an error) */
    const float *gcn_w, *gcn_bias;           /* packed operands of csk_gcn_stage_f32                       */
    const int32_t *ell_src;
    const float *ell_val;
    const float *tcn_w, *tcn_w_res, *tcn_bias; /* packed operands of csk_tcn_step_f32                      */
    float *y_ring;                           /* [y_slots][c_out][P]                                        */
    float *out_ring;                         /* [out_slots][c_out][P]; input history of the next layer     */
    float *tcn_partial;                      /* [partial_emits * tcn_ksplit][c_out][P] or NULL (tcn_ksplit == 1): raw partial sums
                                              * of the emissions ONE launch of this layer produces (may be shared by all
                                              * layers: launches are stream-ordered); a cycle whose launch would emit more
                                              * fails with an error instead of overrunning it */
    /* adaptive graph conv (CoAGCN, models/coa_gcn/coa_gcn.py:11-14); agcn_inter == 0: plain GraphConvolution.  Otherwise the
     * layer's adjacency is computed per skeleton frame by csk_agcn_embed_attention_f32 (per-frame form) into agcn_adj
     * ([agcn_adj_frames * skeletons][3][V][V] floats, may be shared by all layers: launches are stream-ordered) and ell_src /
     * ell_w / ell_cnt describe the dense pattern (ell_w = V, ell_cnt = {V, V, V}); ell_val is not used. */
    int32_t agcn_inter, agcn_pad_;
    const float *agcn_w_pairs, *agcn_b_pairs, *agcn_a_sum;
    float *agcn_adj;
} csk_co_layer;

typedef struct csk_co_plan csk_co_plan;

/* xin0: [xin0_slots][C][P] input ring of layer 0.  xin0_slots = CSK_CO_IN_SLOTS(max_cycle) fixes the largest cycle the plan
 * accepts (max_cycle = xin0_slots - 4, at most CSK_CO_MAX_CYCLE); every layer's rings are checked against it (max_in above
 * with max_cycle in the place of CSK_CO_MAX_CYCLE): a slab for 4-frame cycles is 19 % smaller than one for 8-frame cycles.
 * pool_ring: [pool_size][N][feat_c]; pooled: [N][feat_c]. */
csk_co_plan *csk_co_plan_create(int n_layers, const csk_co_layer *layers, float *xin0, int xin0_slots, int N, int C, int V, int M,
                                int64_t P, const float *bn_scale, const float *bn_shift, int classes,
                                const float *fc_w, const float *fc_b, int pool_size, int pool_padding,
                                float *pool_ring, float *pooled);
void csk_co_plan_destroy(csk_co_plan *plan);
/* swap in refolded weights (same geometry and state rings); counters and state are kept */
int csk_co_plan_update_weights(csk_co_plan *plan, int n_layers, const csk_co_layer *layers, const float *bn_scale,
                               const float *bn_shift, const float *fc_w, const float *fc_b);
/* forget all counters (the caller zeroes the slab): clean_state(), models/base.py:161-164 */
void csk_co_plan_reset(csk_co_plan *plan);
/* read (set = 0) or write (set = 1) the plan's counters: buf = {frames, features, then (received, emitted) per layer},
 * n = 2 + 2*n_layers.  forward_step(x, update_state=False) (models/base.py:183-185) = read, cycle, write back: a
 * step only overwrites ring slots whose content is older than any window, so the counters are the whole state. */
int csk_co_plan_counters(csk_co_plan *plan, int64_t *buf, int n, int set);
/* 1 (default): blocks that qualify advance a 4-frame cycle with one csk_co_block_step_f32 launch instead of a
 * csk_gcn_stage_f32 + csk_tcn_step_f32 pair (bit-identical results); 0: always the two-launch form. */
int csk_co_plan_set_fusion(csk_co_plan *plan, int enable);
/* Advance by r = 1..CSK_CO_MAX_CYCLE frames, frames[i] = (N, C, V, M) device pointers.  On return
 * *last_slot / *n_feat describe the last layer's emissions of this cycle (slot of the first, count) and
 * *n_logits how many predictions were written to `logits` ([CSK_CO_MAX_CYCLE][N][classes], slice j = prediction j). */
int csk_co_plan_cycle(csk_co_plan *plan, const float *const *frames, int r, float *logits, int *last_slot,
                      int *n_feat, int *n_logits, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CSKEL_H */
