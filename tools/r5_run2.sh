cd "$(dirname "$0")/.."
V=$PWD/softgnss-python_amd/lib/variants
for v in fp1 wp10; do
  echo "== $v"
  SGX_LIB=$V/libsgx_$v.so SGX_TRK_PROFILE=1 timeout 300 python tools/step_profile.py 4000 2>&1 | grep "^step\|waveprof\|fineprof" | sort -u | head -40
done
