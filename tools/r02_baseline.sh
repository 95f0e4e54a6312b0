#!/bin/bash
# Round-2 baseline on the GPU box: parity suite, then rocprofv3 passes of the wide-row kernel and the 1M x 768 config.
# usage: gpurun -- 'bash tools/r02_baseline.sh'
set -o pipefail
export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/r02_base
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
for d in 4096 1024; do
  rocprofv3 --kernel-trace --stats -d $O/wide_d${d}_trace -o t --output-format csv -- python3 tools/generic_perf.py $d 2000000 > $O/wide_d${d}.log 2>&1 || { tail $O/wide_d${d}.log; exit 1; }
  rocprofv3 --pmc FETCH_SIZE -d $O/wide_d${d}_fetch -o t --output-format csv -- python3 tools/generic_perf.py $d 2000000 > $O/wide_d${d}_fetch.log 2>&1 || exit 1
  rocprofv3 --pmc WRITE_SIZE -d $O/wide_d${d}_write -o t --output-format csv -- python3 tools/generic_perf.py $d 2000000 > $O/wide_d${d}_write.log 2>&1 || exit 1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $O/wide_d${d}_l2 -o t --output-format csv -- python3 tools/generic_perf.py $d 2000000 > $O/wide_d${d}_l2.log 2>&1 || exit 1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE -d $O/wide_d${d}_sq -o t --output-format csv -- python3 tools/generic_perf.py $d 2000000 > $O/wide_d${d}_sq.log 2>&1 || echo "sq pass failed"
  cat $O/wide_d${d}.log
done
rocprofv3 --kernel-trace --stats -d $O/c2_trace -o t --output-format csv -- python3 tools/quickperf.py 1000000 768 256 > $O/c2.log 2>&1 || exit 1
cat $O/c2.log
ls $O
