#!/bin/bash
# Instruction-mix counters of the two tracking kernels (separate --pmc passes, kernel trace only).
# Usage (GPU box): bash tools/pmc_valu.sh   -> gpurun_out/pmc_valu.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_valu_*
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES" \
           "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_CVT" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_FLAT SQ_INSTS_VALU_FLOPS_FP64" \
           "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES"; do
    i=$((i+1))
    rocprofv3 --pmc $set --kernel-trace -d gpurun_out/pmc_valu_$i -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --concurrent 0 --many-channels 2048 --many-ms 200 > /dev/null 2> gpurun_out/pmc_valu_$i.err || echo "pass $i failed"
done
python3 - <<'PY' | tee gpurun_out/pmc_valu.txt
import glob, sqlite3
for db in sorted(glob.glob("gpurun_out/pmc_valu_*/*/*_results.db")):
    c = sqlite3.connect(db)
    try:
        rows = list(c.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection "
                              "where kernel_name like 'trk_kernel%' group by kernel_name, counter_name order by 1, 2"))
    except Exception as e:
        print(db, "no counters:", e); continue
    for k, n, v, cnt in rows:
        print("%-16s %-28s sum %.6g over %d dispatches" % (k.split('(')[0], n, v, cnt))
PY
