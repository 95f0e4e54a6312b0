#!/bin/bash
set -o pipefail
export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/r03_rs2; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?
tail -6 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
for mode in segments per-source; do
  python tools/config34.py feb4rag 10 $mode > $O/config4_$mode.json 2> $O/config4_$mode.err || { tail $O/config4_$mode.err; exit 1; }
  python -c "import json,sys; j=json.load(open('$O/config4_$mode.json')); print('$mode feb4rag', j['median_ms_per_batch'], j['frac_of_8TBps'])"
  RR_WIDE_RS=0 python tools/config34.py feb4rag 10 $mode > $O/config4_${mode}_rs0.json 2> $O/config4_$mode.err || { tail $O/config4_$mode.err; exit 1; }
  python -c "import json,sys; j=json.load(open('$O/config4_${mode}_rs0.json')); print('$mode feb4rag RR_WIDE_RS=0', j['median_ms_per_batch'], j['frac_of_8TBps'])"
done
timeout -k 10 200 python tools/fuzz_parity.py 601 300 > $O/fuzz.log 2>&1; tail -1 $O/fuzz.log | cut -c1-200
timeout -k 10 300 python tools/fuzz_parity.py 602 500 segments > $O/fuzz_seg.log 2>&1; tail -1 $O/fuzz_seg.log | cut -c1-200
