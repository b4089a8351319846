#!/usr/bin/env python
"""One screen of a bench.py line: python tools/bench_summary.py <file with the JSON line>"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.1f %s, step %.2f us, kernel %.2f us, frac %.4f, e2e %.4f" % (
    d["value"], d["unit"], 1e3 * d["ms_per_step"], 1e3 * d["roofline"]["kernel_ms"], d["roofline"]["frac"],
    d.get("roofline_e2e", {}).get("frac", float("nan"))))
for k in ("predict_roofline", "device_resident_adam_loop", "host_driven_adam_loop", "reference_stream", "gp_samples", "full_elcbo", "cpu_baseline"):
    if k in d:
        v = d[k]
        keep = {kk: vv for kk, vv in v.items() if not isinstance(vv, str) or len(vv) < 40} if isinstance(v, dict) else v
        print(k, json.dumps(keep)[:500])
