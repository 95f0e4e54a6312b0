#!/bin/bash
# GPU box: first measurement pass of round 3 — configs 3 / 4 per source vs segmented, service throughput, list-rerank latency
set -o pipefail
export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/r03_m1
mkdir -p $O
python tools/rerank_latency.py > $O/rerank_latency.json 2> $O/rerank.err || { tail $O/rerank.err; exit 1; }
cat $O/rerank_latency.json
for mode in segments per-source; do
  python tools/config34.py feb4rag 10 $mode > $O/config4_$mode.json 2> $O/config4_$mode.err || { tail $O/config4_$mode.err; exit 1; }
  python -c "import json,sys; j=json.load(open('$O/config4_$mode.json')); print('$mode feb4rag', j['median_ms_per_batch'], j['frac_of_8TBps'])"
done
for mode in segments per-source; do
  python tools/config34.py medrag 10 $mode > $O/config3_$mode.json 2> $O/config3_$mode.err || { tail $O/config3_$mode.err; exit 1; }
  python -c "import json,sys; j=json.load(open('$O/config3_$mode.json')); print('$mode medrag', j['median_ms_per_batch'], j['frac_of_8TBps'])"
done
python tools/shape_bench.py 1000000 768 > $O/shape_1000000_768.json 2>$O/shape.err || { tail $O/shape.err; exit 1; }
cat $O/shape_1000000_768.json
python tools/service_bench.py 10000000 0.2 0.5 2 > $O/service_throughput.json 2> $O/service.err || { tail $O/service.err; exit 1; }
cat $O/service_throughput.json
