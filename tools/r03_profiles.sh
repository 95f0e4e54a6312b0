#!/bin/bash
# GPU box: everything profiles/r03 is made of (final code of the round).  usage: gpurun -- 'bash tools/r03_profiles.sh'
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
export PYTHONPATH=.
O=gpurun_out/r03_prof
mkdir -p $O
VER=$(python -c "from ragroute_amd._lib import lib; print(lib().rr_version())")
# 1. headline: plain run, then the same command under the kernel trace
python bench.py --steps 30 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err || { tail $O/bench_n1.err; exit 1; }
rocprofv3 --kernel-trace --stats -d $O/bench_trace -o t --output-format csv -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --sustained-seconds 0 > $O/bench_n1_under_rocprof.json 2> $O/bench_prof.err || { tail $O/bench_prof.err; exit 1; }
echo "bench traced"
# 2. HBM traffic of the scan launches (separate PMC passes, no trace domains)
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o t --output-format csv -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --sustained-seconds 0 > $O/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o t --output-format csv -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --sustained-seconds 0 > $O/pmc_write.log 2>&1 || exit 1
F=$(find $O/pmc_fetch -name '*counter_collection.csv' | head -1); W=$(find $O/pmc_write -name '*counter_collection.csv' | head -1)
python tools/pmc_traffic.py $F $W 9 $O/traffic.json 10000000 768 256 32 fp16 $VER || exit 1
echo "pmc done"
# 3. other shapes: bench-style lines, three of them also under the kernel trace
for shape in "1000000 768" "4000000 1024" "2000000 4096" "2000000 2048" "10000000 768 1"; do
  tag=$(echo $shape | tr ' ' '_')
  python tools/shape_bench.py $shape > $O/shape_$tag.json 2> $O/shape_$tag.err || { tail $O/shape_$tag.err; exit 1; }
  cat $O/shape_$tag.json
done
for shape in "1000000 768" "4000000 1024" "2000000 4096"; do
  tag=$(echo $shape | tr ' ' '_')
  rocprofv3 --kernel-trace --stats -d $O/trace_$tag -o t --output-format csv -- python3 tools/shape_bench.py $shape > $O/shape_${tag}_under_rocprof.json 2> $O/trace_$tag.err || exit 1
done
echo "shapes done"
# 4. two-rank rehearsals of both scaling modes on the one device (gloo: RCCL refuses two ranks on one GPU)
RR_BENCH_BACKEND=gloo RR_BENCH_ONE_DEVICE=1 python bench.py --gpus 2 --steps 10 --warmup 2 --rows 2000000 --sustained-seconds 0 > $O/bench_gloo2_weak.json 2> $O/bench_w2.err || { tail $O/bench_w2.err; exit 1; }
RR_BENCH_BACKEND=gloo RR_BENCH_ONE_DEVICE=1 python bench.py --gpus 2 --steps 10 --warmup 2 --rows 1000000 --scaling strong --sustained-seconds 0 > $O/bench_gloo2_strong.json 2> $O/bench_s2.err || { tail $O/bench_s2.err; exit 1; }
# 5. config 5 on one GPU
python tools/config5.py > $O/config5.log 2>&1 && tail -1 $O/config5.log > $O/config5_80M_bf16_k100.json
tail -1 $O/config5.log
cat $O/bench_n1.json
