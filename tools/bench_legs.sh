#!/bin/bash
# (diagnosis) the headline step, the many-channel leg and the from-file leg of bench.py, key figures only.  GPU box.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/legs
python bench.py --no-cpu-baseline --concurrent 0 --no-config4 "$@" > gpurun_out/legs/bench.json 2> gpurun_out/legs/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/legs/bench.json").read().strip().splitlines()[-1])
print("step %.3f ms  kernel %.3f ms  x_realtime %.1f" % (d["ms_per_step"], d["track_kernel_ms"], d["x_realtime"]))
m = d.get("roofline_many_channels")
if m:
    print({k: m.get(k) for k in ("frac", "kernel_ms", "kernel_ms_min", "kernel_ms_all", "kernel_ms_grouped_offsets", "error")})
f = d.get("from_file")
if f:
    print({k: v for k, v in f.items() if k not in ("workload", "note")})
PY
