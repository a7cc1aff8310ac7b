#!/bin/bash
# (round 3 diagnosis) issue / wait counters of the two correlation kernels of the acquisition (separate --pmc passes,
# kernel trace only).  GPU box: bash tools/pmc_acq_sq.sh -> gpurun_out/pmc_acq_sq.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_acq_sq_*
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT" \
           "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
    i=$((i+1))
    rocprofv3 --pmc $set --kernel-trace -d gpurun_out/pmc_acq_sq_$i -- python3 tools/acq_once.py 3 > /dev/null 2> gpurun_out/pmc_acq_sq_$i.err || echo "pass $i failed"
done
python3 - <<'PY' | tee gpurun_out/pmc_acq_sq.txt
import glob, sqlite3
for db in sorted(glob.glob("gpurun_out/pmc_acq_sq_*/*/*_results.db")):
    c = sqlite3.connect(db)
    try:
        rows = list(c.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection "
                              "where kernel_name like '%fft4_%' group by kernel_name, counter_name order by 1, 2"))
    except Exception as e:
        print(db, "no counters:", e); continue
    for k, n, v, cnt in rows:
        print("%-60s %-24s sum %.6g over %d dispatches" % (k.split('(')[0][-60:], n, v, cnt))
PY
