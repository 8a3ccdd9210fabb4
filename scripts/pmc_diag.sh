#!/bin/bash
# stall-attribution counter passes for the dominant kernels (counters only with --kernel-trace)
ulimit -c 0
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp
i=0
for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM_RD SQ_BUSY_CYCLES SQ_WAVES" \
            "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
            "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" \
            "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$R/gpurun_out/diag_$i" -o diag -- python "$R/scripts/pmc_diag.py" > "$R/gpurun_out/diag_$i.log" 2>&1
  echo "diag pass $i exit $?"
done
cd "$R"
find gpurun_out -name "*kernel_trace*" -size +8M -delete
python scripts/pmc_diag.py --report gpurun_out/diag_* > gpurun_out/diag_report.txt 2>&1
cat gpurun_out/diag_report.txt
