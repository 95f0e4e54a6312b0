#!/bin/bash
# GPU box: parity of the wide-row kernel variants, then A/B timing (RR_WIDE_PD=0: round-1 kernel, 2/3: deeper pipeline)
set -o pipefail
export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/r02_wide
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_flat_search_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; rc=$?
tail -5 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
for pd in ${PDS:-0 2}; do
  for d in 4096 1024 2048; do
    RR_WIDE_PD=$pd timeout -k 10 300 python tools/generic_perf.py $d ${ROWS:-2000000} > $O/perf_pd${pd}_d${d}.log 2>&1 || { tail $O/perf_pd${pd}_d${d}.log; exit 1; }
    echo "pd=$pd $(grep 'nq=' $O/perf_pd${pd}_d${d}.log | tr '\n' ' ')"
  done
done
