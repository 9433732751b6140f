#!/bin/bash
# Evidence of a round on the GPU box (run through gpurun from the repo root): tests, bench, rocprofv3 kernel stats of
# the same bench command, the two PMC traffic passes and one SQ pass.  Usage: bash tools/collect_round.sh r02 v1
set -o pipefail
R=${1:-r02}; V=${2:-v1}
ROOT=$PWD; OUT=$ROOT/gpurun_out/$R$V; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $OUT/gputests.log 2>&1 || { tail -20 $OUT/gputests.log; exit 1; }
tail -2 $OUT/gputests.log
timeout -k 10 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o b -- python3 $ROOT/bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> /dev/null || exit 1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_f -o f --output-format csv -- python3 $ROOT/bench.py --steps 1 --warmup 1 --layers 2 --no-cpu-baseline > /dev/null 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_w -o w --output-format csv -- python3 $ROOT/bench.py --steps 1 --warmup 1 --layers 2 --no-cpu-baseline > /dev/null 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --kernel-include-regex "gemm_fast|gemm_w8|fused_select|fused_bucket|topk_rows_kernel|decode_fwd_multi|rows_lds" --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES -d $OUT/pmc_sq -o s --output-format csv -- python3 $ROOT/bench.py --steps 2 --warmup 1 --layers 4 --no-cpu-baseline > /dev/null 2>&1 || exit 1
cd $ROOT
python3 tools/pmc_traffic.py $(find $OUT/pmc_f -name "*counter_collection.csv" | head -1) $(find $OUT/pmc_w -name "*counter_collection.csv" | head -1) $OUT/pmc_traffic.json > $OUT/pmc_traffic.txt
python3 tools/pmc_sq_summary.py $(find $OUT/pmc_sq -name "*counter_collection.csv" | head -1) > $OUT/pmc_sq.txt
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/stats $OUT/pmc_f $OUT/pmc_w $OUT/pmc_sq
python3 tools/show_bench.py $OUT/bench.json
python3 tools/kstats.py $OUT 6
