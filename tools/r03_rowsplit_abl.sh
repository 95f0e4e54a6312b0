#!/bin/bash
# GPU box: price a row-split 8-wave layout whose query fragments come from LDS (timing-only ablations of the 8-wave kernel,
# RR_DEV_VARIANTS libraries): _a64 baseline (insertion path shut), _a72 half the A-fragment reads, _a328 half the A reads + 8 extra
# reads per step (= 16 A + 8 B per wave: the new layout's LDS load; its vector-memory bytes equal today's).  Scores wrong by design.
export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/r03_rowsplit; mkdir -p $O
for rep in 1 2; do
for shape in "4000000 1024" "2000000 2048"; do
  for L in _a64 _a72 _a328; do
    f=$O/shape_$(echo $shape | tr ' ' x)$L.json
    RR_WIDE_WAVES=8 RR_LIB_OVERRIDE=ragroute_amd/libragroute_hip$L.so timeout -k 10 200 python tools/shape_bench.py $shape 256 10 fp16 20 > $f 2> $f.err || { tail -3 $f.err; continue; }
    python - "$f" "$shape lib=$L" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = j["roofline"]
print(sys.argv[2], "scan frac", r["frac"], "avg_launch_ms", r["avg_launch_ms"], "b2b_ms", j["back_to_back_ms"])
PY
  done
done
done
