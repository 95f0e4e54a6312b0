#!/bin/bash
# GPU box: final-code refresh of the shape lines, kernel stats and configs 3 / 4 for profiles/r03
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
export PYTHONPATH=.
O=gpurun_out/r03_final; mkdir -p $O
for shape in "1000000 768" "4000000 1024" "2000000 4096" "2000000 2048"; do
  tag=$(echo $shape | tr ' ' '_')
  python tools/shape_bench.py $shape > $O/shape_$tag.json 2> $O/shape_$tag.err || { tail $O/shape_$tag.err; exit 1; }
  cat $O/shape_$tag.json | cut -c1-420
done
for shape in "4000000 1024" "2000000 4096"; do
  tag=$(echo $shape | tr ' ' '_')
  rocprofv3 --kernel-trace --stats -d $O/trace_$tag -o t --output-format csv -- python3 tools/shape_bench.py $shape > $O/shape_${tag}_under_rocprof.json 2> $O/trace_$tag.err || exit 1
done
for mode in segments per-source; do
  python tools/config34.py feb4rag 10 $mode > $O/config4_$mode.json 2> $O/config4_$mode.err || { tail $O/config4_$mode.err; exit 1; }
  python -c "import json,sys; j=json.load(open('$O/config4_$mode.json')); print('$mode feb4rag', j['median_ms_per_batch'], j['frac_of_8TBps'])"
done
RR_WIDE_RS=0 python tools/config34.py feb4rag 10 segments > $O/config4_segments_rs0.json 2>/dev/null
python -c "import json,sys; j=json.load(open('$O/config4_segments_rs0.json')); print('segments feb4rag RR_WIDE_RS=0', j['median_ms_per_batch'], j['frac_of_8TBps'])"
python tools/config34.py medrag 10 segments > $O/config3_segments.json 2>/dev/null
python -c "import json,sys; j=json.load(open('$O/config3_segments.json')); print('segments medrag', j['median_ms_per_batch'], j['frac_of_8TBps'])"
python bench.py --steps 30 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err || { tail $O/bench_n1.err; exit 1; }
cut -c1-300 $O/bench_n1.json
