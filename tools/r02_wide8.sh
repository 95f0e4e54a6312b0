#!/bin/bash
# GPU box: parity of the wide-row kernels (8-wave default), then A/B timing RR_WIDE_WAVES=4 vs 8
set -o pipefail
export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/r02_wide8
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_flat_search_gpu.py tests/test_baseline_configs_gpu.py::test_config4_two_ranks_on_one_device tests/test_router_merge_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; rc=$?
tail -5 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
for w in 4 8; do
  for d in 4096 1024 2048 1536; do
    RR_WIDE_WAVES=$w timeout -k 10 300 python tools/generic_perf.py $d ${ROWS:-2000000} > $O/perf_w${w}_d${d}.log 2>&1 || { tail $O/perf_w${w}_d${d}.log; exit 1; }
    echo "waves=$w $(grep 'nq=' $O/perf_w${w}_d${d}.log | tr '\n' ' ')"
  done
done
