#!/bin/bash
# GPU box: bench.py --workload at N=1 and a two-process rehearsal on the one device (gloo exchange); plan identity check
export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/r04_workload; mkdir -p $O
for ds in feb4rag medrag; do
  timeout -k 10 300 python bench.py --workload $ds --steps 10 --warmup 3 --sustained-seconds 0 > $O/bench_${ds}_n1.json 2> $O/bench_${ds}_n1.err || tail -5 $O/bench_${ds}_n1.err
  RR_BENCH_BACKEND=gloo RR_BENCH_ONE_DEVICE=1 timeout -k 10 400 python bench.py --workload $ds --gpus 2 --steps 10 --warmup 3 --sustained-seconds 0 > $O/bench_${ds}_gloo2.json 2> $O/bench_${ds}_gloo2.err || tail -5 $O/bench_${ds}_gloo2.err
done
RR_BENCH_BACKEND=gloo RR_BENCH_ONE_DEVICE=1 timeout -k 10 400 python bench.py --workload feb4rag --placement whole --gpus 2 --steps 10 --warmup 3 --sustained-seconds 0 > $O/bench_feb4rag_gloo2_whole.json 2> $O/bench_feb4rag_gloo2_whole.err || tail -5 $O/bench_feb4rag_gloo2_whole.err
RR_BENCH_BACKEND=gloo RR_BENCH_ONE_DEVICE=1 timeout -k 10 400 python bench.py --gpus 2 --steps 10 --warmup 3 --sustained-seconds 0 --rows 2000000 > $O/bench_headline_gloo2.json 2> $O/bench_headline_gloo2.err || tail -5 $O/bench_headline_gloo2.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04_workload/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], j["value"], j["ms_per_step"], j["result_checksum"], j["per_rank_local_ms"]["ranks"], j["per_rank_scan_ms"]["ranks"], j["roofline"]["frac"])
    except Exception as e: print(f, "failed", e)
PY
for ds in feb4rag medrag; do
  timeout -k 10 400 python tools/config34.py $ds 10 --plan 8 --with-one > $O/plan8_$ds.json 2> $O/plan8_$ds.err || tail -5 $O/plan8_$ds.err
  python -c "
import json; j=json.loads(open('$O/plan8_$ds.json').read().strip().splitlines()[-1]); print('$ds', j['max_over_mean'], j['predicted_speedup_at_G'], j['merged_G_ranks_equal_one_gpu'])"
done
