#!/bin/bash
# GPU box: full parity suite (incl. the round-3 tests), then the service-level throughput sweep
set -o pipefail
export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/r03_m2
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?
tail -15 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
python tools/service_bench.py 10000000 0.2 0.5 2 > $O/service_throughput.json 2> $O/service.err || { tail $O/service.err; exit 1; }
cat $O/service_throughput.json
