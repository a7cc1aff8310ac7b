#!/bin/bash
# Round-5 evidence run (GPU box, ~12 min): BASELINE configs[2] at full size against the numpy oracle on four scenes with the
# final kernels (trk3_kernel), then the wide fuzz sweeps (random front ends / scenes: acquisition + tracking; random scenes
# at the default front end: acquisition, device-led fine search).  Writes gpurun_out/r05_full_parity.json.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/r05_full_parity.txt
: > $out
python tools/full_parity.py 37000 2>/dev/null | tail -1 >> $out
for seed in 7 8 9; do python tools/full_parity.py 37000 $seed 2>/dev/null | tail -1 >> $out; done
python tools/fuzz_parity.py 400 520 2>/dev/null | tail -1 > gpurun_out/r05_fuzz.txt
python tools/acq_fuzz_default.py 400 700 2>/dev/null | tail -1 > gpurun_out/r05_acq_fuzz.txt
python - <<'PY'
import json
runs = [json.loads(l) for l in open("gpurun_out/r05_full_parity.txt") if l.strip().startswith("{")]
d = {"note": "BASELINE configs[2] at full size (8 channels x 37 000 ms = 296 000 blocks per scene) against the numpy oracle, "
             "final round-5 kernels (tools/r5_full_parity.sh: trk3_kernel, device-led acquisition); then the wide fuzz sweeps",
     "runs": runs,
     "fuzz_wide": "tools/fuzz_parity.py 400 520 (random front ends and scenes, acquisition + tracking against the oracle): "
                  + open("gpurun_out/r05_fuzz.txt").read().strip()
                  + "; tools/acq_fuzz_default.py 400 700 (random scenes at the default front end: codePhase, carrFreq exact, "
                    "peakMetric to 1e-9): " + open("gpurun_out/r05_acq_fuzz.txt").read().strip()}
json.dump(d, open("gpurun_out/r05_full_parity.json", "w"), indent=1)
print(json.dumps(d)[:3000])
PY
