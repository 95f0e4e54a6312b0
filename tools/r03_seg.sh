#!/bin/bash
# GPU box: segmented-search parity, then configs 3 / 4 per source vs segmented
set -o pipefail
export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/r03_seg
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_segments_gpu.py tests/test_flat_search_gpu.py tests/test_fuzz_gpu.py tests/test_guard_pages_gpu.py -m gpu -x -q > $O/pytest_new.log 2>&1; rc=$?
tail -25 $O/pytest_new.log
[ $rc -ne 0 ] && exit $rc
for mode in segments per-source; do
  python tools/config34.py medrag 10 $mode > $O/config3_$mode.json 2> $O/config3_$mode.err || { tail $O/config3_$mode.err; exit 1; }
  python -c "import json,sys; j=json.load(open('$O/config3_$mode.json')); print('$mode medrag', j['median_ms_per_batch'], j['frac_of_8TBps'])"
done
for mode in segments per-source; do
  python tools/config34.py feb4rag 10 $mode > $O/config4_$mode.json 2> $O/config4_$mode.err || { tail $O/config4_$mode.err; exit 1; }
  python -c "import json,sys; j=json.load(open('$O/config4_$mode.json')); print('$mode feb4rag', j['median_ms_per_batch'], j['frac_of_8TBps'])"
done
python tools/shape_bench.py 10000000 768 > $O/shape_10M.json 2>$O/shape.err || { tail $O/shape.err; exit 1; }
cat $O/shape_10M.json
