#!/bin/bash
# (round 5) the three full-run tests on the default scene for library variants: bash tools/r5_fullcheck.sh default NAME ...
cd "$(dirname "$0")/.."
for v in "$@"; do
  if [ "$v" = default ]; then unset SGX_LIB; else export SGX_LIB=$PWD/softgnss-python_amd/lib/variants/libsgx_$v.so; fi
  echo "== $v: $(timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k 'one_workgroup or config3 or config5 or split_variants' 2>&1 | tail -1)"
done
