#!/bin/bash
# GPU box: what would hiding the wide-row kernel's per-group epilogue gain?  Timing-only ablation libraries (RR_DEV_VARIANTS builds):
# _abl64 = insertion path shut (baseline), _abl = the same without any epilogue.  Scores are wrong by design; only times are read.
export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/r03_epi; mkdir -p $O
for shape in "4000000 1024" "2000000 2048"; do
  for L in _abl64 _abl; do
    f=$O/shape_$(echo $shape | tr ' ' x)$L.json
    RR_LIB_OVERRIDE=ragroute_amd/libragroute_hip$L.so timeout -k 10 200 python tools/shape_bench.py $shape 256 10 fp16 20 > $f 2> $f.err || { tail -3 $f.err; continue; }
    python - "$f" "$shape lib=$L" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = j["roofline"]
print(sys.argv[2], "scan frac", r["frac"], "avg_launch_ms", r["avg_launch_ms"], "b2b_ms", j["back_to_back_ms"])
PY
  done
done
