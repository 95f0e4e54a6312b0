#!/bin/bash
# GPU box: config 2 (1M x 768, 256 queries, k = 32): bootstrap sample size A/B (development library) and the per-search kernel timeline
export TMPDIR=/tmp PYTHONPATH=.
R=$GRAFT_REPO_ROOT
OUT=c2_sample SHAPES="1000000 768" K=32 ITERS=50 REPS=3 ENVS="RR_SAMPLE_ROWS=8192;RR_SAMPLE_ROWS=4096;RR_SAMPLE_ROWS=2048" bash tools/collect.sh ab _dev
O=gpurun_out/c2_trace; mkdir -p $O
(cd /tmp && rocprofv3 --kernel-trace -d $R/$O/trace -o t --output-format csv -- python3 $R/tools/shape_bench.py 1000000 768 256 32 fp16 30 > $R/$O/shape.json 2> $R/$O/trace.err)
f=$(find $O/trace -name '*kernel_trace.csv' | head -1); echo $f
python - $f <<'PY'
import csv, sys, collections
rows=[r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
print([r["Kernel_Name"][:40] for r in rows[-24:]])
PY
for per in 9 10 11 8; do python tools/trace_gaps.py $f $per 20 > $O/gaps.json 2> $O/gaps.err && { echo per=$per; cat $O/gaps.json; break; }; done
