#!/bin/bash
# (runs on the GPU box through gpurun: rocprofv3 kernel stats of tools/bench_fused.py for each "lib:CLT_GEMM_DEBUG" argument)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
run() {  # lib dbg
  export CLT_LIB=$R/clt_training_amd/$1 CLT_GEMM_DEBUG=$2
  rm -rf /tmp/abl_x
  timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl_x -o p -- python3 $R/tools/bench_fused.py > /dev/null 2>&1 || exit 1
  f=$(find /tmp/abl_x -name "*kernel_stats.csv" | head -1)
  echo -n "$1 dbg=$2: "; python3 - $f <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'gemm_fast' in r['Name'] and 'Accum' in r['Name'] and r['Calls']=='24':
        print(r['Name'][31:50], round(float(r['AverageNs'])/1e3,1))
PY
}
for a in "$@"; do run ${a%%:*} ${a##*:}; done
