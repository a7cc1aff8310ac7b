#!/bin/bash
# (diagnosis) duration of the two correlation kernels for side-by-side builds of the library: lib/libsgx_<name>.so. GPU box.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for name in "$@"; do
  rm -rf gpurun_out/prof_var
  export SGX_LIB=$GRAFT_REPO_ROOT/softgnss-python_amd/lib/libsgx_$name.so
  rocprofv3 --kernel-trace -d gpurun_out/prof_var -- python3 tools/acq_once.py 4 > gpurun_out/var_$name.log 2>&1
  tail -1 gpurun_out/var_$name.log
  python3 - <<PY
import glob, sqlite3
db = glob.glob("gpurun_out/prof_var/*/*_results.db")[0]
c = sqlite3.connect(db)
out = []
for pat in ("%fft4_cols_kernel%4, 1>%", "%fft4_rows_kernel%1, 2%"):
    r = list(c.execute("select duration from kernels where name like ? order by start", (pat,)))
    big = sorted(x[0] for x in r if x[0] > 100000)
    out.append(min(big) / 1e3 if big else -1)
print("$name  cols<1> %7.1f us   rows<2> %7.1f us" % (out[0], out[1]))
PY
done
