#!/usr/bin/env bash
set -u
timeout 1500 python -m pytest tests/test_gpu_clip_parity.py tests/test_gpu_precision_modes.py tests/test_gpu_edge_cases.py tests/test_gpu_bench_smoke.py -x -q 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/profile_bench_stats.sh r06 > gpurun_out/profile_r06_bench.log 2>&1; tail -3 gpurun_out/profile_r06_bench.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_line.json 2> gpurun_out/r06_bench_line.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r06_bench_line.json").read().strip().splitlines()[-1])
print("value",d["value"],"frac",d["roofline"]["frac"],"ms",d["ms_per_step"], d["roofline_config"]["frac"], d["roofline_config"]["frac_alg"])
o=d["costgcn_online"]; print("online",o["value"],o["roofline"]["frac"],o["roofline_config"]["frac"],o["roofline_config"]["frac_alg"],o["throughput_mode"], o["state_slab_GB_per_gpu"], o["split_k_scratch_GB_per_gpu"])
a=d["agcn_kinetics"]; print("agcn clip",a["agcn_clip"]["value"],a["agcn_clip"]["roofline_config"]["frac"],"coagcn",a["coagcn_online"]["value"],a["coagcn_online"]["roofline_config"]["frac"], a["coagcn_online"]["state_slab_GB_per_gpu"])
print("latency",[ (x["streams"],x["ms_per_frame_p50"],x["pipelined_ms_per_frame"]) for x in o["latency"]["per_stream_count"]])
print("clip latency",[(b["batch"],b["default"]["graph_ms_p50"],b["latency_mode"]["graph_ms_p50"]) for b in d["clip_latency"]["per_batch"]])
print("bf16x3", d["clip_bf16x3"]["value"] if "clip_bf16x3" in d else None, "config5", d["config5"]["value"], d["config5"]["roofline"]["frac"])
print("cpu", d["cpu_baseline"]["value"], o["cpu_baseline"]["value"])
PY
