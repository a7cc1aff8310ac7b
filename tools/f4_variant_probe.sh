#!/bin/bash
# (diagnosis) duration of the two correlation kernels for side-by-side builds of the library
# (tools/build_variant.sh: lib/variants/libsgx_<name>.so; "default" = the shipped build). GPU box.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for name in "$@"; do
  rm -rf gpurun_out/prof_var
  if [ "$name" = default ]; then unset SGX_LIB; else export SGX_LIB=$GRAFT_REPO_ROOT/softgnss-python_amd/lib/variants/libsgx_$name.so; fi
  rocprofv3 --kernel-trace -d gpurun_out/prof_var -- python3 tools/acq_once.py 4 > gpurun_out/var_$name.log 2>&1
  python3 - <<PY
import glob, sqlite3
db = glob.glob("gpurun_out/prof_var/*/*_results.db")[0]
c = sqlite3.connect(db)
out = []
for pat in ("%fft4_cols%", "%fft4_rows%"):
    r = list(c.execute("select duration from kernels where name like ? order by start", (pat,)))
    big = sorted(x[0] for x in r if x[0] > 30000)
    out.append((big[0] / 1e3, big[len(big) // 2] / 1e3) if big else (-1, -1))
print("%-12s cols min %7.1f med %7.1f us   rows min %7.1f med %7.1f us" % ("$name", out[0][0], out[0][1], out[1][0], out[1][1]))
PY
done
