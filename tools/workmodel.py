"""Algorithmic work of the ST-GCN / CoST-GCN forward path (SURVEY.md 8d accounting), shared by bench.py and the
profile summarisers.  Pure Python, no torch.

MAC = one multiply-add = 2 FLOP.  Bytes = fused-block minimum: block input read once + block output written once
(clip); new frame in + post-GCN frame to the ring + 8 ring frames re-read + residual FIFO r/w + output (step).
`dense` aggregation MACs (3 * V per input element, 3.8 % of the clip total) are what SURVEY 8d credits; the sparse
kernel executes only the non-zeros (`agg_nnz` per element: 6 of 75 for NTU-25), both totals are reported.
"""

PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: dense fp32 MFMA (v_mfma_f32_32x32x2_f32)
PEAK_HBM_TBS = 8.0               # MI355X_MICROARCH.md: HBM3E

LAYERS = [(3, 64, 1, False), (64, 64, 1, True), (64, 64, 1, True), (64, 64, 1, True), (64, 128, 2, True),
          (128, 128, 1, True), (128, 128, 1, True), (128, 256, 2, True), (256, 256, 1, True), (256, 256, 1, True)]


def layer_table(c_in=3):
    return [(c_in if i == 0 else ci, co, s, r) for i, (ci, co, s, r) in enumerate(LAYERS)]


def per_position_macs(ci, co, s, res, V=25, agg_nnz=6, adaptive=False):
    """MACs per INPUT position (gcn part) and per OUTPUT position (tcn part) of one block.
    adaptive (A-GCN, models/a_gcn/a_gcn.py:48-69): + the six 1x1 embedding convs a_i / b_i (C_in -> C_out/4 each:
    6 * ci * co/4), + the attention logits (per sample and subset V x V x (inter * T) MACs = 3 * V * co/4 per position;
    the same count per position with T = 1 in step mode), and the aggregation is DENSE (3 * V per input element is
    executed, not merely credited)."""
    gcn_conv = 3 * ci * co
    gcn_res = ci * co if ci != co else 0
    agg_dense = 3 * V * ci
    agg_sparse = agg_dense if adaptive else agg_nnz * ci
    embed = 6 * ci * (co // 4) if adaptive else 0
    attn = 3 * V * (co // 4) if adaptive else 0
    tcn = 9 * co * co
    blk_res = ci * co if (res and (ci != co or s != 1)) else 0
    return dict(gcn=gcn_conv + gcn_res + embed + attn, agg_dense=agg_dense, agg_sparse=agg_sparse, tcn=tcn + blk_res,
                embed=embed, attn=attn)


def clip_layers(T=300, V=25, c_in=3, adaptive=False, agg_nnz=None):
    """Per skeleton sequence: list of dicts (gcn_macs, tcn_macs, agg_dense, agg_sparse, bytes) per layer.
    agg_nnz = non-zeros of A per input element over the three subsets (NTU-25: 6 of 75; OpenPose-18: 5 of 54)."""
    agg_nnz = (6 if V == 25 else 5) if agg_nnz is None else agg_nnz
    out, t = [], T
    for (ci, co, s, res) in layer_table(c_in):
        t_out = (t - 1) // s + 1
        m = per_position_macs(ci, co, s, res, V, agg_nnz, adaptive)
        out.append(dict(ci=ci, co=co, stride=s, t_in=t, t_out=t_out,
                        gcn_macs=m["gcn"] * t * V, agg_dense=m["agg_dense"] * t * V, agg_sparse=m["agg_sparse"] * t * V,
                        embed_macs=m["embed"] * t * V, attn_macs=m["attn"] * t * V,
                        tcn_macs=m["tcn"] * t_out * V, bytes=4 * V * (ci * t + co * t_out)))
        t = t_out
    return out


def clip_totals(n_seq, T=300, V=25, c_in=3, adaptive=False):
    """(flops_alg [SURVEY: dense aggregation credited], flops_exec [sparse], bytes_alg) for n_seq skeleton sequences."""
    ls = clip_layers(T, V, c_in, adaptive)
    base = sum(l["gcn_macs"] + l["tcn_macs"] for l in ls)
    return (2 * n_seq * (base + sum(l["agg_dense"] for l in ls)), 2 * n_seq * (base + sum(l["agg_sparse"] for l in ls)),
            n_seq * sum(l["bytes"] for l in ls))


def step_layers(frames=4, V=25, c_in=3, adaptive=False, agg_nnz=None):
    """Per skeleton and per cycle of `frames` input frames (a multiple of the stack's stride 4): list of dicts per
    layer with frames_in / emissions and MACs."""
    agg_nnz = (6 if V == 25 else 5) if agg_nnz is None else agg_nnz
    out, rate = [], frames
    for (ci, co, s, res) in layer_table(c_in):
        m = per_position_macs(ci, co, s, res, V, agg_nnz, adaptive)
        emis = rate // s
        mode_bytes = 0 if not res else (2 * ci)      # residual FIFO write + read per input frame
        by = 4 * V * (rate * (ci + co + mode_bytes) + emis * (8 * co + co))
        out.append(dict(ci=ci, co=co, stride=s, frames_in=rate, emissions=emis, gcn_macs=m["gcn"] * rate * V,
                        embed_macs=m["embed"] * rate * V, attn_macs=m["attn"] * rate * V,
                        agg_dense=m["agg_dense"] * rate * V, agg_sparse=m["agg_sparse"] * rate * V,
                        tcn_macs=m["tcn"] * emis * V, bytes=by))
        rate = emis
    return out


def step_totals(n_skel, frames=4, V=25, c_in=3, adaptive=False):
    """(flops_alg, flops_exec, bytes_alg) of one cycle of `frames` frames for n_skel skeletons.  The byte figure is the
    frame-rate-weighted step model; SURVEY 8d's table quotes the unweighted per-layer sum (825.9 KB per skeleton-frame),
    an upper bound of it."""
    ls = step_layers(frames, V, c_in, adaptive)
    base = sum(l["gcn_macs"] + l["tcn_macs"] for l in ls)
    return (2 * n_skel * (base + sum(l["agg_dense"] for l in ls)), 2 * n_skel * (base + sum(l["agg_sparse"] for l in ls)),
            n_skel * sum(l["bytes"] for l in ls))


def roofline_config(flops_alg, bytes_alg, t_measured_s, flops_exec=None, n_gpus=1):
    """SURVEY 8d / BASELINE.md 3: t_MFMA, t_HBM, t_roof = max, fraction = t_roof / t_measured.
    `flops_alg`, `flops_exec`, `bytes_alg` are the work of the WHOLE JOB in the measured time (all `n_gpus` ranks) and are
    priced against the job's peak, `n_gpus` x one MI355X (157.3 TFLOP/s fp32 MFMA, 8 TB/s HBM): a weak-scaling run that
    scales perfectly keeps the N = 1 fraction at every N, and no fraction can exceed 1.
    `frac` (the headline) prices only the FLOPs the kernels EXECUTE (`flops_executed`: the sparse GCN kernel skips the
    zeros of the skeleton adjacency); `frac_alg` is the same with SURVEY 8d's accounting, which credits the dense
    3 * V aggregation MACs per element (3.5 % of the ST-GCN total that is never executed)."""
    if n_gpus < 1:
        raise ValueError("roofline_config: n_gpus must be >= 1")
    fe = flops_alg if flops_exec is None else flops_exec
    peak_f = PEAK_F32_MFMA_TFLOPS * 1e12 * n_gpus
    peak_b = PEAK_HBM_TBS * 1e12 * n_gpus
    t_mfma_alg = flops_alg / peak_f
    t_mfma = fe / peak_f
    t_hbm = bytes_alg / peak_b
    t_roof = max(t_mfma, t_hbm)
    return dict(flops_alg=flops_alg, flops_executed=fe, bytes_alg=bytes_alg, n_gpus=n_gpus,
                peak_tflops=round(PEAK_F32_MFMA_TFLOPS * n_gpus, 1), peak_hbm_tbs=round(PEAK_HBM_TBS * n_gpus, 1),
                t_mfma_ms=round(t_mfma * 1e3, 4),
                t_hbm_ms=round(t_hbm * 1e3, 4), t_roof_ms=round(t_roof * 1e3, 4), t_measured_ms=round(t_measured_s * 1e3, 4),
                bound="mfma" if t_mfma >= t_hbm else "hbm", frac=round(t_roof / t_measured_s, 6),
                frac_executed=round(t_roof / t_measured_s, 6),
                frac_alg=round(max(t_mfma_alg, t_hbm) / t_measured_s, 6),   # (6 digits: toy-size runs sharing a GPU must not round to 0)
                achieved_tflops=round(fe / t_measured_s / 1e12, 2),
                achieved_tflops_per_gpu=round(fe / t_measured_s / 1e12 / n_gpus, 2),
                achieved_hbm_alg_tbs=round(bytes_alg / t_measured_s / 1e12, 3))


if __name__ == "__main__":
    for tag, ad in (("ST-GCN", False), ("A-GCN", True)):
        fa, fe, by = clip_totals(128, V=18, adaptive=ad)
        print(f"{tag} Kinetics shape, batch 64: {fa / 1e12:.3f} TFLOP alg ({fe / 1e12:.3f} executed), {by / 1e9:.2f} GB")
        fa, fe, by = step_totals(2048, 4, V=18, adaptive=ad)
        print(f"Co{tag} Kinetics shape, 1024 streams, 4-frame cycle: {fa / 1e9:.1f} GFLOP alg ({fe / 1e9:.1f} executed), {by / 1e9:.3f} GB")
    fa, fe, by = clip_totals(512)
    print(f"clip batch 256: {fa / 1e12:.3f} TFLOP alg ({fe / 1e12:.3f} executed), {by / 1e9:.2f} GB")
    fa, fe, by = step_totals(2048, 4)
    print(f"online 1024 streams, 4-frame cycle: {fa / 1e9:.1f} GFLOP alg ({fe / 1e9:.1f} executed), {by / 1e9:.3f} GB;"
          f" per frame-step {fa / 4e9:.1f} GFLOP")
