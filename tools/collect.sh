#!/bin/bash
# GPU box: the ONE collection script (replaces the per-round tools/r02_*.sh / r03_*.sh / r04/*.sh one-offs).
#
#   gpurun --timeout N -- 'bash tools/collect.sh <task> [args]'        output under gpurun_out/<OUT:-task>/
#
# tasks
#   suite                 whole GPU suite, smoke(), bench N=1, two-rank gloo rehearsals (weak / strong) on the one device
#   tests FILES...        the given test files only (-m gpu -x -q)
#   profiles              everything a round's profiles/rNN is made of: bench line, the same command under rocprofv3 --kernel-trace
#                         --stats, FETCH_SIZE / WRITE_SIZE passes -> traffic.json, SHAPES lines (+ traces), config 5
#   shapes                bench-style lines for SHAPES (default: config 2, the wide rows, one query), K / ITERS / BATCH apply
#   ab LIB...             same-device A/B of library suffixes (ragroute_amd/libragroute_hip<suffix>.so via RR_LIB_OVERRIDE; "main" = the
#                         product library) on SHAPES, alternated REPS times.  ENVS="A=1;B=2 C=3" adds one leg per ';'-separated
#                         environment (development libraries read RR_* tuning variables, product libraries ignore them)
#   abtree DIR            bench.py of this tree against the tree in DIR (its own bench.py and library, e.g. `git archive <rev>` + build),
#                         alternated REPS times, then rocprofv3 --kernel-trace --stats of each
#   pmc LIB SETS...       counter passes (one rocprofv3 --pmc run per quoted SET) over tools/shape_bench.py SHAPE with library suffix LIB
#   workload              bench.py --workload feb4rag|medrag at N=1, two ranks on the one device (gloo), sliced and whole placement;
#                         tools/config34.py --plan 8 --with-one for both federations
#   service [ROWS WINDOWS...]   tools/service_bench.py
#   fuzz SEED COUNT [segments|big]   tools/fuzz_parity.py
#   probe CYCLES          tools/remap_probe.py (stale-input root-cause probe)
#   rswiden               the row-split kernel's widened dispatch against round 3's limits (development library _dev, one leg per old limit;
#                         profiles/r04/rowsplit_widening.json)
#   config2               config 2: bootstrap sample size A/B (development library) + per-search kernel timeline (tools/trace_gaps.py;
#                         profiles/r04/config2_floor.json)
# variables: OUT, SHAPES ("rows dim;rows dim ..."), K, ITERS, BATCH, REPS, ENVS, STEPS
export TMPDIR=/tmp PYTHONPATH=.
R=${GRAFT_REPO_ROOT:-$(pwd)}
task=$1; shift
O=gpurun_out/${OUT:-$task}; mkdir -p $O; O_BASE=${OUT:-$task}
SHAPES=${SHAPES:-"1000000 768;4000000 1024;2000000 2048;2000000 4096;10000000 768 1"}
REPS=${REPS:-2}; STEPS=${STEPS:-30}

last() { python - "$@" <<'PY'
import json, sys
for f in sys.argv[1:]:
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "no result line:", e); continue
    r = j.get("roofline", {})
    print(f.split("/")[-1], "value", j.get("value"), "ms", j.get("ms_per_step", j.get("back_to_back_ms")), "scan frac", r.get("frac"), "avg launch ms", r.get("avg_launch_ms"),
          "sustained", (r.get("sustained") or {}).get("frac"), "b2b frac", j.get("back_to_back_frac_of_8TBps"), j.get("result_checksum", ""), (j.get("per_rank_local_ms") or {}).get("ranks", ""))
PY
}
lib_of() { [ "$1" = main ] && echo ragroute_amd/libragroute_hip.so || echo ragroute_amd/libragroute_hip$1.so; }

case $task in
suite)
  timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; rc=$?; tail -5 $O/pytest_gpu.log; [ $rc -ne 0 ] && exit $rc
  timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail $O/smoke.log; exit 1; }; tail -1 $O/smoke.log
  timeout -k 10 300 python bench.py --steps $STEPS --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err || { tail $O/bench_n1.err; exit 1; }
  RR_BENCH_BACKEND=gloo RR_BENCH_ONE_DEVICE=1 timeout -k 10 300 python bench.py --gpus 2 --steps 10 --warmup 2 --rows 2000000 --sustained-seconds 0 > $O/bench_gloo2_weak.json 2> $O/bench_w2.err || { tail $O/bench_w2.err; exit 1; }
  RR_BENCH_BACKEND=gloo RR_BENCH_ONE_DEVICE=1 timeout -k 10 300 python bench.py --gpus 2 --steps 10 --warmup 2 --rows 1000000 --scaling strong --sustained-seconds 0 > $O/bench_gloo2_strong.json 2> $O/bench_s2.err || { tail $O/bench_s2.err; exit 1; }
  last $O/bench_n1.json $O/bench_gloo2_weak.json $O/bench_gloo2_strong.json ;;
tests)
  timeout -k 10 1000 python -m pytest "$@" -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -12 $O/pytest.log; exit $rc ;;
profiles)
  VER=$(python -c "from ragroute_amd._lib import lib; print(lib().rr_version())")
  python bench.py --steps $STEPS --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err || { tail $O/bench_n1.err; exit 1; }
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $R/$O/bench_trace -o t --output-format csv -- python3 $R/bench.py --steps $STEPS --warmup 5 --no-cpu-baseline --sustained-seconds 0 > $R/$O/bench_n1_under_rocprof.json 2> $R/$O/bench_prof.err) || { tail $O/bench_prof.err; exit 1; }
  (cd /tmp && rocprofv3 --pmc FETCH_SIZE -d $R/$O/pmc_fetch -o t --output-format csv -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --sustained-seconds 0 > $R/$O/pmc_fetch.log 2>&1) || exit 1
  (cd /tmp && rocprofv3 --pmc WRITE_SIZE -d $R/$O/pmc_write -o t --output-format csv -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --sustained-seconds 0 > $R/$O/pmc_write.log 2>&1) || exit 1
  F=$(find $O/pmc_fetch -name '*counter_collection.csv' | head -1); W=$(find $O/pmc_write -name '*counter_collection.csv' | head -1)
  python tools/pmc_traffic.py $F $W 9 $O/traffic.json 10000000 768 256 32 fp16 $VER || exit 1
  IFS=';' read -ra SH <<< "$SHAPES"
  for shape in "${SH[@]}"; do
    tag=$(echo $shape | tr ' ' '_')
    python tools/shape_bench.py $shape > $O/shape_$tag.json 2> $O/shape_$tag.err || { tail $O/shape_$tag.err; exit 1; }
    [ "${TRACE_SHAPES:-1}" = 1 ] && (cd /tmp && rocprofv3 --kernel-trace --stats -d $R/$O/trace_$tag -o t --output-format csv -- python3 $R/tools/shape_bench.py $shape > $R/$O/shape_${tag}_under_rocprof.json 2> $R/$O/trace_$tag.err)
  done
  python tools/config5.py > $O/config5.log 2>&1 && tail -1 $O/config5.log > $O/config5_80M_bf16_k100.json
  last $O/bench_n1.json $O/shape_*.json ;;
shapes)
  IFS=';' read -ra SH <<< "$SHAPES"
  for shape in "${SH[@]}"; do
    set -- $shape; tag=$(echo $shape | tr ' ' '_')
    python tools/shape_bench.py $1 $2 ${3:-${BATCH:-256}} ${K:-32} ${DTYPE:-fp16} ${ITERS:-30} ${METRIC:-ip} > $O/shape_${tag}_k${K:-32}_${METRIC:-ip}.json 2> $O/shape_$tag.err || { tail -3 $O/shape_$tag.err; continue; }
    last $O/shape_${tag}_k${K:-32}_${METRIC:-ip}.json
  done ;;
ab)
  IFS=';' read -ra SH <<< "$SHAPES"; IFS=';' read -ra EV <<< "${ENVS:-;}"; [ ${#EV[@]} -eq 0 ] && EV=("")
  for shape in "${SH[@]}"; do for rep in $(seq $REPS); do for L in "$@"; do for ei in "${!EV[@]}"; do
    set -- "$@"; f=$O/shape_$(echo $shape | tr ' ' 'x')_${L}_e${ei}_$rep.json
    env ${EV[$ei]} RR_LIB_OVERRIDE=$(lib_of $L) timeout -k 10 250 python tools/shape_bench.py $shape ${BATCH:-256} ${K:-32} fp16 ${ITERS:-30} ${METRIC:-ip} > $f 2> $f.err || { tail -3 $f.err; continue; }
    echo -n "[$shape | lib=$L | ${EV[$ei]} | rep $rep] "; last $f
  done; done; done; done ;;
abtree)
  DIR=$1
  for rep in $(seq $REPS); do
    timeout -k 10 200 python bench.py --steps $STEPS --warmup 5 --no-cpu-baseline > $O/head_$rep.json 2> $O/head_$rep.err || tail -3 $O/head_$rep.err
    (cd $DIR && PYTHONPATH=. timeout -k 10 200 python bench.py --steps $STEPS --warmup 5 --no-cpu-baseline > $R/$O/other_$rep.json 2> $R/$O/other_$rep.err) || tail -3 $O/other_$rep.err
  done
  last $O/head_*.json $O/other_*.json
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $R/$O/prof_head -o head --output-format csv -- python3 $R/bench.py --steps $STEPS --warmup 5 --no-cpu-baseline --sustained-seconds 0 > $R/$O/prof_head.json 2> $R/$O/prof_head.err)
  (cd $R/$DIR && PYTHONPATH=. rocprofv3 --kernel-trace --stats -d $R/$O/prof_other -o other --output-format csv -- python3 bench.py --steps $STEPS --warmup 5 --no-cpu-baseline --sustained-seconds 0 > $R/$O/prof_other.json 2> $R/$O/prof_other.err)
  for f in $(find $O -name '*kernel_stats.csv'); do echo $f; head -4 $f | cut -c1-160; done ;;
pmc)
  L=$1; shift; i=0
  for set in "$@"; do
    i=$((i + 1))
    (cd /tmp && env ${ENVS} RR_LIB_OVERRIDE=$R/$(lib_of $L) timeout -k 10 200 rocprofv3 --pmc $set -d $R/$O/p$i -o t --output-format csv -- python3 $R/tools/shape_bench.py ${SHAPE:-4000000 1024} ${BATCH:-256} ${K:-10} fp16 4 > $R/$O/p$i.log 2>&1); echo "pass $i ($set) rc=$?"
  done
  python3 - $O <<'PY'
import collections, csv, glob, sys
acc, n = collections.defaultdict(float), collections.defaultdict(int)
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "flat_scan" in r["Kernel_Name"]:
            key = (r["Kernel_Name"].split("(")[0][-60:], r["Counter_Name"])
            acc[key] += float(r["Counter_Value"]); n[key] += 1
for (k, c), v in sorted(acc.items()):
    print(k, c, round(v / n[(k, c)], 1), "x", n[(k, c)])
PY
  ;;
workload)
  for ds in feb4rag medrag; do
    timeout -k 10 300 python bench.py --workload $ds --steps 10 --warmup 3 --sustained-seconds 0 > $O/bench_${ds}_n1.json 2> $O/bench_${ds}_n1.err || tail -5 $O/bench_${ds}_n1.err
    RR_BENCH_BACKEND=gloo RR_BENCH_ONE_DEVICE=1 timeout -k 10 400 python bench.py --workload $ds --gpus 2 --steps 10 --warmup 3 --sustained-seconds 0 > $O/bench_${ds}_gloo2.json 2> $O/bench_${ds}_gloo2.err || tail -5 $O/bench_${ds}_gloo2.err
    RR_BENCH_BACKEND=gloo RR_BENCH_ONE_DEVICE=1 timeout -k 10 400 python bench.py --workload $ds --placement whole --gpus 2 --steps 10 --warmup 3 --sustained-seconds 0 > $O/bench_${ds}_gloo2_whole.json 2> $O/bench_${ds}_gloo2_whole.err || tail -5 $O/bench_${ds}_gloo2_whole.err
    timeout -k 10 400 python tools/config34.py $ds 10 --plan 8 --with-one > $O/plan8_$ds.json 2> $O/plan8_$ds.err || tail -5 $O/plan8_$ds.err
    python -c "
import json; j = json.loads(open('$O/plan8_$ds.json').read().strip().splitlines()[-1]); print('$ds plan 8: max/mean', j['max_over_mean'], 'max ms', j['max_ms'], 'one GPU ms', j.get('one_gpu_ms'), 'speedup', j.get('predicted_speedup_at_G'), j.get('merged_G_ranks_equal_one_gpu'))"
  done
  last $O/bench_*.json ;;
service)
  timeout -k 10 700 python tools/service_bench.py "$@" > $O/service.json 2> $O/service.err; grep -v amdgpu.ids $O/service.err | cut -c1-330 ;;
fuzz)
  timeout -k 10 1000 python tools/fuzz_parity.py "$@" > $O/fuzz_$(echo "$@" | tr ' ' '_').log 2>&1; rc=$?; tail -1 $O/fuzz_$(echo "$@" | tr ' ' '_').log | cut -c1-300; exit $rc ;;
rswiden)
  OUT=$O_BASE/rs_k300 SHAPES="2000000 2048" K=300 ITERS=20 ENVS="RR_WIDE_RS_MAXK=128;RR_WIDE_RS_MAXK=1024" bash tools/collect.sh ab _dev
  OUT=$O_BASE/rs_l2 SHAPES="4000000 1024;2000000 4096" K=10 METRIC=l2 ITERS=20 ENVS="RR_WIDE_RS_L2=0;RR_WIDE_RS_L2=1" bash tools/collect.sh ab _dev
  for q in 193 200 208; do
    OUT=$O_BASE/rs_q$q SHAPES="4000000 1024;2000000 2048;2000000 4096" K=10 BATCH=$q ITERS=20 ENVS="RR_WIDE_RS_MINQ=209;RR_WIDE_RS_MINQ=193" bash tools/collect.sh ab _dev
  done ;;
config2)
  OUT=$O_BASE/c2_sample SHAPES="1000000 768" K=32 ITERS=50 REPS=3 ENVS="RR_SAMPLE_ROWS=8192;RR_SAMPLE_ROWS=4096;RR_SAMPLE_ROWS=2048" bash tools/collect.sh ab _dev
  (cd /tmp && rocprofv3 --kernel-trace -d $R/$O/trace -o t --output-format csv -- python3 $R/tools/shape_bench.py 1000000 768 256 32 fp16 30 > $R/$O/shape.json 2> $R/$O/trace.err)
  python tools/trace_gaps.py $(find $O/trace -name '*kernel_trace.csv' | head -1) 9 20 | tee $O/gaps.json ;;
probe)
  timeout -k 10 900 python tools/remap_probe.py "$@" > $O/remap_probe.log 2>&1; rc=$?; tail -3 $O/remap_probe.log | cut -c1-1500; exit $rc ;;
*) echo "unknown task $task"; exit 2 ;;
esac
