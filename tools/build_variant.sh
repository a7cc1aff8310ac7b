#!/bin/bash
# (diagnosis) side-by-side build of the library with one translation unit recompiled with extra flags:
#   bash tools/build_variant.sh NAME sgx_fft.hip "-DF4W_NOSTORE"   -> softgnss-python_amd/lib/variants/libsgx_NAME.so
# (run a variant with SGX_LIB=<path>; lib/variants is git-ignored and travels to the GPU box)
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; flags=$3
lib=softgnss-python_amd/lib
mkdir -p $lib/variants
# (a variant of the speculative tracking kernel must leave the polls' reserved registers alone: tools/check_trk3_regs.py)
if [ "$src" = sgx_trk3.hip ] && [ -z "$SGX_SKIP_REG_GATE" ]; then python3 tools/check_trk3_regs.py $flags || exit 1; fi
/opt/rocm/bin/hipcc $flags --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -x hip -I include -I softgnss-python_amd/csrc \
    -Wno-unused-result -Wno-unused-value -c softgnss-python_amd/csrc/$src -o $lib/variants/$name.o
objs=$(ls $lib/obj/*.o | grep -v "/${4:-$src}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $lib/variants/libsgx_$name.so $objs $lib/variants/$name.o -ldl
rm -f $lib/variants/$name.o
echo built $lib/variants/libsgx_$name.so
