#!/bin/bash
# (round 3 diagnosis) XCD-aware tile order of the columns kernel against dispatch order (SGX_ACQ_XCD=0): parity, kernel
# durations, bytes fetched.  GPU box: bash tools/acq_xcd_probe.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/acq_xcd
mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "acq" > $out/pytest.log 2>&1
echo "pytest rc $?"; tail -2 $out/pytest.log
for v in 1 0; do
  export SGX_ACQ_XCD=$v
  rm -rf gpurun_out/prof_var gpurun_out/prof_varf
  rocprofv3 --kernel-trace -d gpurun_out/prof_var -- python3 tools/acq_once.py 6 > $out/var_$v.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/prof_varf -- python3 tools/acq_once.py 3 > /dev/null 2>&1
  python3 - <<PY
import glob, sqlite3
db = glob.glob("gpurun_out/prof_var/*/*_results.db")[0]
c = sqlite3.connect(db)
for pat in ("%fft4_cols%", "%fft4_rows%"):
    r = list(c.execute("select name, duration from kernels where name like ? order by start", (pat,)))
    big = sorted(x[1] for x in r if x[1] > 100000)
    print("XCD=$v  %-14s n=%d min %7.1f med %7.1f us" % (pat, len(big), big[0] / 1e3, big[len(big)//2] / 1e3))
db = glob.glob("gpurun_out/prof_varf/*/*_results.db")[0]
c = sqlite3.connect(db)
rows = list(c.execute("select kernel_name, sum(value), count(*) from counters_collection where kernel_name like '%fft4_%' group by kernel_name"))
for k, v, n in rows:
    print("XCD=$v  FETCH_SIZE %-44s %.1f MB raw per call (x2 for wide reads)" % (k.split('(')[0][-44:], v / 3 / 1024.0))
PY
  python3 tools/acq_once.py 6 2>&1 | grep acquire_ms; python3 tools/acq_once4.py 4 2>&1 | tail -1
done
