#!/usr/bin/env python3
"""gpurun_out/parity_report.jsonl (one line per tests/helpers.check_parity call of a `pytest -m gpu` run) ->
profiles/<tag>_parity_report.json: a summary per precision mode (checks, worst error, largest reference magnitude, tolerance)
plus every record, sorted by error.  usage: python tools/make_parity_report.py [jsonl] [tag]"""
import collections
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "parity_report.jsonl")
tag = sys.argv[2] if len(sys.argv) > 2 else "r04"
recs = [json.loads(ln) for ln in open(src) if ln.strip()]
modes = collections.defaultdict(list)
for r in recs:
    modes[r.get("mode", "f32")].append(r)
summary = {m: {"checks": len(v), "worst_max_abs_err": max(r["max_abs_err"] for r in v),
               "largest_absmax_ref": max(r["absmax_ref"] for r in v), "tol": max(r["tol"] for r in v),
               "tests": len({r["test"] for r in v})} for m, v in modes.items()}
out = {"note": "one record per parity assertion of `python -m pytest tests -m gpu` on an MI355X (tests/helpers.check_parity: "
               "|got - want| <= tol ABSOLUTE, |want| <= 32 asserted); mode f32 = default exact-fp32 path, bf16x3 = opt-in split arithmetic",
       "summary": summary, "records": sorted(recs, key=lambda r: -r["max_abs_err"])}
dst = os.path.join(ROOT, "profiles", f"{tag}_parity_report.json")
with open(dst, "w") as f:
    f.write(json.dumps(out, indent=0))
print(dst, json.dumps(summary))
