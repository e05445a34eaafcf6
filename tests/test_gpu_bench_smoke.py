"""bench.py end to end at toy sizes: one JSON line with the contract's keys (metric, value, unit, n_gpus, steps,
warmup, ms_per_step, higher_is_better, scaling, vs_baseline, dtype, data, config.workload, roofline, cpu_baseline)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], cwd=ROOT, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("workload", ["clip", "step", "both"])
def test_bench_line_contract(workload):
    d = _run("--workload", workload, "--batch", "4", "--streams", "8", "--steps", "2", "--warmup", "1", "--step-cycles", "2",
             "--stream-shards", "2", "--cpu-budget", "6", "--config4-batch", "2", "--config4-streams", "6",
             "--cpu-budget-config4", "4", "--latency-frames", "12", "--config5-batch", "6", "--clip-latency-forwards", "6",
             "--kernel-cycles", "2")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and d["ms_per_step"] > 0
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 157.3 and 0 < r["frac"] < 1
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and c["sample"]
    rc = d["roofline_config"]
    assert rc["frac"] == rc["frac_executed"] and rc["frac_alg"] >= rc["frac"] and rc["flops_executed"] <= rc["flops_alg"]
    assert d["ranks_seen"] == 1 and d["collective_backend"] is None
    if workload in ("clip", "both"):          # configs[4]'s per-GPU shard (1024 clips in a real run) beside the headline, also at N = 1
        assert d["config5"]["clips_per_gpu"] == 6 and d["config5"]["global_batch"] == 6 and d["config5"]["value"] > 0
        assert "no collective" in d["config5"]["workload"] and 0 < d["config5"]["roofline_config"]["frac"] < 1
    if workload in ("clip", "both"):          # small-batch clip latency: batch 1 and 8, eager and hipGraph, default and latency mode
        cl = d["clip_latency"]
        assert [r["batch"] for r in cl["per_batch"]] == [1, 8] and cl["cpu_oracle_batch1_ms"] > 0
        for r in cl["per_batch"]:
            for mode in ("default", "latency_mode"):
                e = r[mode]
                assert 0 < e["graph_ms_p50"] <= e["graph_ms_p99"] and 0 < e["eager_ms_p50"] <= e["eager_ms_p99"] and e["pipelined_ms"] > 0
                assert e["max_abs_logit_diff_vs_default"] < 1e-4
            assert r["default"]["split_k"] == 0 and r["latency_mode"]["split_k"] == 4 and 0 < r["roofline_config_frac_best_graph_p50"] < 1
    if workload in ("clip", "both"):          # the opt-in precision mode is reported under its own key, never as the headline
        b3 = d["clip_bf16x3"]
        assert b3["value"] > 0 and "bf16x3" in b3["dtype"] and b3["max_abs_logit_diff_vs_f32"] < 1e-4 and d["dtype"] == "f32"
    if workload == "both":        # BASELINE configs[3] legs carry their own whole-config roofline and CPU baseline
        k = d["agcn_kinetics"]
        for leg, unit in (("agcn_clip", "clips/s"), ("coagcn_online", "frames/s")):
            assert k[leg]["value"] > 0 and k[leg]["unit"] == unit
            assert k[leg]["roofline_config"]["bound"] == "mfma" and 0 < k[leg]["roofline_config"]["frac"] < 1
            assert k[leg]["cpu_baseline"]["kind"] == "port" and k[leg]["cpu_baseline"]["value"] > 0
        b3 = k["agcn_clip"]["bf16x3"]                     # opt-in mode of the A-GCN clip leg: its own key
        assert b3["value"] > 0 and "bf16x3" in b3["dtype"] and b3["max_abs_logit_diff_vs_f32"] < 1e-4
        assert d["costgcn_online"]["roofline_config"]["frac"] > 0
    if workload in ("step", "both"):      # few-stream latency leg (the reference's batch-1 protocol): 1 and 16 streams
        lat = (d if workload == "step" else d["costgcn_online"]).get("latency") or d.get("costgcn_online", {}).get("latency")
        assert lat and [e["streams"] for e in lat["per_stream_count"]] == [1, 16]
        for e in lat["per_stream_count"]:
            assert 0 < e["ms_per_frame_p50"] <= e["ms_per_frame_p99"] and e["launches_per_frame"] > 10 and e["predictions"] == 3
        assert lat["cpu_oracle_ms_per_frame_one_stream"] > 0
