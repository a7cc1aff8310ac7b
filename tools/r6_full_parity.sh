#!/bin/bash
# Round-6 evidence run (GPU box, ~25 min): config 3 in full against the oracle on the default + 16 random scenes with the
# first divergence diagnosed (tools/r6_parity_rate.py), then the wide fuzz sweeps over 600 + 300 seeds.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python3 tools/r6_parity_rate.py ${1:-16} ${2:-100} 2>gpurun_out/r06_parity_rate.err > gpurun_out/r06_full_parity_${2:-100}.jsonl
python3 tools/fuzz_parity.py 1000 1600 2>/dev/null | tail -1 > gpurun_out/r06_fuzz.txt
python3 tools/acq_fuzz_default.py 1000 1300 2>/dev/null | tail -1 > gpurun_out/r06_acq_fuzz.txt
tail -3 gpurun_out/r06_parity_rate.err; cat gpurun_out/r06_full_parity_${2:-100}.jsonl | cut -c1-300; cat gpurun_out/r06_fuzz.txt gpurun_out/r06_acq_fuzz.txt
