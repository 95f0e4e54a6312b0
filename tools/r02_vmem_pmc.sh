#!/bin/bash
# GPU box: where does the vector-memory path of the wide-row step stall?  TA / TCP / TD / SQ-VMEM counters of the scan launches
# (separate --pmc passes of at most two counters per TA / TCP / TD block - more "exceeds the capabilities of the hardware";
# shape_bench with 4 iterations), for d = 4096 (4-wave wide-row kernel) against d = 768.
set -o pipefail
cd /tmp; export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_vmem; mkdir -p $O
for shape in ${SHAPES:-"2000000 4096" "10000000 768"}; do
  tag=$(echo $shape | tr ' ' '_')
  i=0
  for set in "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" \
             "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum GRBM_GUI_ACTIVE" \
             "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
             "TCP_TCR_TCP_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
             "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE" \
             "TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE" \
             "SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" \
             "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES"; do
    i=$((i + 1))
    timeout -k 10 150 rocprofv3 --pmc $set -d $O/${tag}_p$i -o t --output-format csv -- python3 $R/tools/shape_bench.py $shape 256 32 fp16 4 > $O/${tag}_p$i.log 2>&1
    echo "$tag pass $i rc=$? $(grep -c flat_scan $O/${tag}_p$i/t_counter_collection.csv 2>/dev/null) rows"
  done
done
