#!/bin/bash
# (diagnosis) config 4 (32 PRNs x 10 ms non-coherent) acquisition time for side-by-side builds: lib/libsgx_<name>.so. GPU box.
cd "$GRAFT_REPO_ROOT"
for name in "$@"; do
  echo -n "$name  "
  SGX_LIB=$GRAFT_REPO_ROOT/softgnss-python_amd/lib/libsgx_$name.so python3 tools/acq_once4.py 5 2>&1 | tail -1
done
