#!/bin/bash
# (round 3 diagnosis) the one-wave columns kernel against the four-wave one (SGX_ACQ_COLS4=1): parity tests, then the
# durations of the correlation kernels and the time of a call.  GPU box: bash tools/acq_cols_probe.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/acq_cols
mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "acq" > $out/pytest.log 2>&1
echo "pytest rc $?"; tail -3 $out/pytest.log
python3 tools/f4_check.py 2>&1 | tail -1
for v in 0 1; do
  export SGX_ACQ_COLS4=$v
  rm -rf gpurun_out/prof_var
  rocprofv3 --kernel-trace -d gpurun_out/prof_var -- python3 tools/acq_once.py 6 > $out/var_$v.log 2>&1
  tail -1 $out/var_$v.log
  python3 - <<PY
import glob, sqlite3
db = glob.glob("gpurun_out/prof_var/*/*_results.db")[0]
c = sqlite3.connect(db)
for pat in ("%fft4_cols%", "%fft4_rows_kernel%"):
    r = list(c.execute("select name, duration from kernels where name like ? order by start", (pat,)))
    big = sorted(x[1] for x in r if x[1] > 100000)
    print("COLS4=$v  %-40s n=%d min %7.1f med %7.1f us" % (pat, len(big), big[0] / 1e3 if big else -1, big[len(big)//2] / 1e3 if big else -1))
PY
  python3 tools/acq_once4.py 4 2>&1 | tail -1
done
