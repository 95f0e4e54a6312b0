#!/bin/bash
export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/r04_service; mkdir -p $O
timeout -k 10 600 python tools/service_bench.py 10000000 0.2 1.0 > $O/service.json 2> $O/service.err; grep -v amdgpu.ids $O/service.err | cut -c1-330
for rep in 1 2; do
  timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $O/head_$rep.json 2> $O/head_$rep.err || tail -3 $O/head_$rep.err
  (cd ab_r02 && PYTHONPATH=. timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > ../$O/r02_$rep.json 2> ../$O/r02_$rep.err) || tail -3 $O/r02_$rep.err
done
python - <<'PY'
import json
for t in ("head_1","r02_1","head_2","r02_2"):
    try:
        j=json.loads(open(f"gpurun_out/r04_service/{t}.json").read().strip().splitlines()[-1]); r=j["roofline"]
        print(t, j["value"], j["ms_per_step"], j["median_ms"], j["p90_ms"], r["frac"], r["avg_launch_ms"], r["sustained"]["frac"], r["sustained"]["avg_launch_ms"])
    except Exception as e: print(t, "failed", e)
PY
