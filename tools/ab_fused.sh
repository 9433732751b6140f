#!/bin/bash
# kernel-level A/B of two builds on tools/bench_fused.py (rocprofv3 kernel stats)
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for tag in new prev; do
  if [ $tag = prev ]; then export CLT_LIB=$R/clt_training_amd/libcltmi355_prev.so; else unset CLT_LIB; fi
  timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o p -- python3 $R/tools/bench_fused.py > $R/gpurun_out/prof_$tag.log 2>&1
  f=$(find $R/gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
  echo "== $tag"; cut -d, -f1-4 $f | head -12 | cut -c1-150
done
