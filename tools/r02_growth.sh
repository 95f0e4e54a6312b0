#!/bin/bash
# GPU box: chunk growth sweep (RR_CHUNK_GROWTH pins the growth factor; unset = the cost-model schedule)
export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/r02_growth; mkdir -p $O
for shape in "1000000 768" "10000000 768" "300000 768" "4000000 1024"; do
  for G in auto 3 5 8 11 16 35 128; do
    f=$O/shape_$(echo $shape | tr ' ' x)_g$G.json
    if [ $G = auto ]; then unset RR_CHUNK_GROWTH; else export RR_CHUNK_GROWTH=$G; fi
    timeout -k 10 200 python tools/shape_bench.py $shape 256 ${K:-32} fp16 30 > $f 2> $f.err || { tail -3 $f.err; continue; }
    python - "$f" "$shape growth=$G" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = j["roofline"]
print(sys.argv[2], "launches", r["scan_launches_per_search"], "scan frac", r["frac"], "b2b_ms", j["back_to_back_ms"], "b2b frac", j["back_to_back_frac_of_8TBps"], "sane", j["sanity_top1"])
PY
  done
done
