#!/bin/bash
# GPU box, round 4 first call: slice-unit parity, same-device A/B of the headline (HEAD vs the round-2 tree in ab_r02/), plan rehearsal
export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/r04_first; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_placement_gpu.py -x -q > $O/pytest_placement.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_placement.log
for rep in 1 2; do
  timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $O/head_$rep.json 2> $O/head_$rep.err || tail -3 $O/head_$rep.err
  (cd ab_r02 && PYTHONPATH=. timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > ../$O/r02_$rep.json 2> ../$O/r02_$rep.err) || tail -3 $O/r02_$rep.err
done
python - <<'PY'
import json
for t in ("head_1","r02_1","head_2","r02_2"):
    try:
        j=json.loads(open(f"gpurun_out/r04_first/{t}.json").read().strip().splitlines()[-1])
        print(t, j["value"], j["ms_per_step"], j["roofline"]["frac"], j["roofline"]["avg_launch_ms"], j["roofline"].get("sustained",{}).get("frac"))
    except Exception as e: print(t, "failed", e)
PY
cd /tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_head -o head -- python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --sustained-seconds 0 > $GRAFT_REPO_ROOT/$O/prof_head.json 2> $GRAFT_REPO_ROOT/$O/prof_head.err
cd $GRAFT_REPO_ROOT/ab_r02 && PYTHONPATH=. rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_r02 -o r02 -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --sustained-seconds 0 > $GRAFT_REPO_ROOT/$O/prof_r02.json 2> $GRAFT_REPO_ROOT/$O/prof_r02.err
cd $GRAFT_REPO_ROOT
find $O -name "*kernel_stats.csv" | while read f; do echo $f; head -6 $f; done
for ds in feb4rag medrag; do
  timeout -k 10 400 python tools/config34.py $ds 10 --plan 8 --with-one > $O/plan8_$ds.json 2> $O/plan8_$ds.err || tail -5 $O/plan8_$ds.err
  python - $O/plan8_$ds.json <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(j["config"]); print("max",j["max_ms"],"mean",j["mean_ms"],"max/mean",j["max_over_mean"],"pred max",j["predicted_max_ms"],"merge",j["merge_of_G_ranks_ms"],"one",j.get("one_gpu_ms"),"speedup",j.get("predicted_speedup_at_G"))
for r in j["ranks"]: print(r["rank"], r["measured_ms"], r["predicted_ms"], r["corpus_GB"], len(r["units"]))
PY
done
