#!/bin/bash
# (round 5) tracking kernel time against the phase of the filter waves' polls (T3_PH1 / T3_PH2, sgx_trk3.hip): variants
# built with tools/build_variant.sh NAME sgx_trk3.hip "-DT3_PH1=a -DT3_PH2=b".  GPU box: bash tools/r5_phase.sh NAME ...
cd "$(dirname "$0")/.."
bash tools/trk_ab.sh 2 default "$@" | sort | awk '{n[$1]++; s[$1]+=$3} END {for (k in n) printf "%-10s track_kernel_ms %.3f\n", k, s[k]/n[k]}' | sort
