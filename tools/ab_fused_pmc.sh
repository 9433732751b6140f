#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for tag in new prev; do
  if [ $tag = prev ]; then export CLT_LIB=$R/clt_training_amd/libcltmi355_prev.so; else unset CLT_LIB; fi
  rm -rf /tmp/pmc_$tag
  timeout -s KILL 150 rocprofv3 --kernel-include-regex "gemm_fast" --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_WAIT_INST_LDS -d /tmp/pmc_$tag -o s --output-format csv -- python3 $R/tools/bench_fused.py > /dev/null 2>&1 || exit 1
  echo "== $tag"; python3 $R/tools/pmc_sq_summary.py $(find /tmp/pmc_$tag -name "*counter_collection.csv" | head -1)
done
