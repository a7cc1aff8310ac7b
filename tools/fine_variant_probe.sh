#!/bin/bash
# (diagnosis) duration of the two fine-search kernels for side-by-side builds of the library: lib/variants/libsgx_<name>.so (tools/build_variant.sh). GPU box.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for name in "$@"; do
  rm -rf gpurun_out/prof_var
  export SGX_LIB=$GRAFT_REPO_ROOT/softgnss-python_amd/lib/variants/libsgx_$name.so
  rocprofv3 --kernel-trace -d gpurun_out/prof_var -- python3 tools/acq_once.py 4 > gpurun_out/var_$name.log 2>&1
  python3 - <<PY
import glob, sqlite3
db = glob.glob("gpurun_out/prof_var/*/*_results.db")[0]
c = sqlite3.connect(db)
out = []
for pat in ("fine_cols_kernel%", "fine_rows_kernel%"):
    r = [x[0] for x in c.execute("select duration from kernels where name like ? order by start", (pat,))]
    out.append(min(r) / 1e3 if r else -1)
print("$name  fine_cols %7.1f us   fine_rows %7.1f us" % (out[0], out[1]))
PY
done
