#!/bin/bash
# GPU box: everything profiles/r02 is made of.  usage: gpurun -- 'bash tools/r02_profiles.sh'
set -o pipefail
export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/r02_prof
mkdir -p $O
VER=$(python -c "from ragroute_amd._lib import lib; print(lib().rr_version())")
# 1. headline: plain run, then the same under the kernel trace
python bench.py --steps 30 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err || { tail $O/bench_n1.err; exit 1; }
rocprofv3 --kernel-trace --stats -d $O/bench_trace -o t --output-format csv -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --sustained-seconds 0 > $O/bench_n1_under_rocprof.json 2> $O/bench_prof.err || { tail $O/bench_prof.err; exit 1; }
# 2. HBM traffic of the scan launches (separate PMC passes)
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o t --output-format csv -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --sustained-seconds 0 > $O/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o t --output-format csv -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --sustained-seconds 0 > $O/pmc_write.log 2>&1 || exit 1
python tools/pmc_traffic.py $O/pmc_fetch/t_counter_collection.csv $O/pmc_write/t_counter_collection.csv 5 $O/traffic.json 10000000 768 256 32 fp16 $VER || exit 1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $O/pmc_sq -o t --output-format csv -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --sustained-seconds 0 > $O/pmc_sq.log 2>&1 || echo "sq pass failed"
# 3. other shapes: bench-style lines + kernel stats
for shape in "1000000 768" "4000000 1024" "2000000 4096" "2000000 2048" "10000000 768 1" "4000000 1024 1" "2000000 4096 1"; do
  tag=$(echo $shape | tr ' ' '_')
  python tools/shape_bench.py $shape > $O/shape_$tag.json 2> $O/shape_$tag.err || { tail $O/shape_$tag.err; exit 1; }
  cat $O/shape_$tag.json
done
for shape in "1000000 768" "4000000 1024" "2000000 4096"; do
  tag=$(echo $shape | tr ' ' '_')
  rocprofv3 --kernel-trace --stats -d $O/trace_$tag -o t --output-format csv -- python3 tools/shape_bench.py $shape > $O/shape_${tag}_under_rocprof.json 2> $O/trace_$tag.err || exit 1
done
# 4. wide-row kernels: HBM bytes, L2 hit / miss, SQ counters
for shape in "4000000 1024" "2000000 4096"; do
  tag=$(echo $shape | tr ' ' '_')
  rocprofv3 --pmc FETCH_SIZE -d $O/wide_fetch_$tag -o t --output-format csv -- python3 tools/shape_bench.py $shape 256 32 fp16 4 > /dev/null 2>&1 || exit 1
  rocprofv3 --pmc WRITE_SIZE -d $O/wide_write_$tag -o t --output-format csv -- python3 tools/shape_bench.py $shape 256 32 fp16 4 > /dev/null 2>&1 || exit 1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $O/wide_l2_$tag -o t --output-format csv -- python3 tools/shape_bench.py $shape 256 32 fp16 4 > /dev/null 2>&1 || exit 1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $O/wide_sq_$tag -o t --output-format csv -- python3 tools/shape_bench.py $shape 256 32 fp16 4 > /dev/null 2>&1 || echo "sq pass failed"
done
# 5. is the wide-row kernel issue-bound or power-bound?  the same counters on the timing-only ablations (results wrong by design)
for v in _abl2 _abl6; do
  [ -f ragroute_amd/libragroute_hip$v.so ] || continue
  RR_WIDE_WAVES=4 RR_LIB_OVERRIDE=ragroute_amd/libragroute_hip$v.so rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/abl_sq$v -o t --output-format csv -- python3 tools/shape_bench.py 2000000 4096 256 32 fp16 4 > /dev/null 2>&1 || echo "abl pass failed"
  RR_WIDE_WAVES=4 RR_LIB_OVERRIDE=ragroute_amd/libragroute_hip$v.so rocprofv3 --kernel-trace --stats -d $O/abl_trace$v -o t --output-format csv -- python3 tools/shape_bench.py 2000000 4096 256 32 fp16 4 > /dev/null 2>&1
done
RR_WIDE_WAVES=4 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/abl_sq_base -o t --output-format csv -- python3 tools/shape_bench.py 2000000 4096 256 32 fp16 4 > /dev/null 2>&1
RR_WIDE_WAVES=4 rocprofv3 --kernel-trace --stats -d $O/abl_trace_base -o t --output-format csv -- python3 tools/shape_bench.py 2000000 4096 256 32 fp16 4 > /dev/null 2>&1
# 6. config 5 on one GPU
python tools/config5.py > $O/config5.log 2>&1 && tail -1 $O/config5.log > $O/config5_80M_bf16_k100.json
tail -1 $O/config5.log
cat $O/bench_n1.json
