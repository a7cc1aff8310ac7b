#!/bin/bash
# Counters of the throughput-mode tracking kernel (separate --pmc passes, kernel trace only), 2048 channels x 500 ms.
# Usage (GPU box): bash tools/pmc_tp.sh   -> gpurun_out/pmc_tp.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_tp_*
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD" \
           "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES"; do
    i=$((i+1))
    rocprofv3 --pmc $set --kernel-trace -d gpurun_out/pmc_tp_$i -- python3 tools/tp_time.py > /dev/null 2> gpurun_out/pmc_tp_$i.err || echo "pass $i failed"
done
python3 - <<'PY' | tee gpurun_out/pmc_tp.txt
import glob, sqlite3
for db in sorted(glob.glob("gpurun_out/pmc_tp_*/*/*_results.db")):
    c = sqlite3.connect(db)
    try:
        rows = list(c.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection "
                              "where kernel_name like 'trk_kernel_tp%' group by kernel_name, counter_name order by 1, 2"))
    except Exception as e:
        print(db, "no counters:", e); continue
    for k, n, v, cnt in rows:
        print("%-16s %-28s sum %.6g over %d dispatches" % (k.split('(')[0], n, v, cnt))
print("(dispatches: 20 ms + 3 x 500 ms of 2048 channels = 2048 x 1520 blocks of 38192 samples; 1023 chips / 64 lanes = 16 wave-iterations per block and wave... )")
PY
