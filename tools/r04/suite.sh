#!/bin/bash
# GPU box: the whole GPU suite + smoke + service throughput + the headline on the rebuilt library
export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/r04_suite; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
for rep in 1 2; do
  timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $O/head_$rep.json 2> $O/head_$rep.err || tail -3 $O/head_$rep.err
  (cd ab_r02 && PYTHONPATH=. timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > ../$O/r02_$rep.json 2> ../$O/r02_$rep.err) || tail -3 $O/r02_$rep.err
done
python - <<'PY'
import json
for t in ("head_1","r02_1","head_2","r02_2"):
    try:
        j=json.loads(open(f"gpurun_out/r04_suite/{t}.json").read().strip().splitlines()[-1])
        print(t, j["value"], j["ms_per_step"], j["roofline"]["frac"], j["roofline"]["avg_launch_ms"], j["roofline"].get("sustained",{}).get("frac"))
    except Exception as e: print(t, "failed", e)
PY
timeout -k 10 500 python tools/service_bench.py 10000000 0.2 1.0 > $O/service.json 2> $O/service.err; tail -25 $O/service.err
