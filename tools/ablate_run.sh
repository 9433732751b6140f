#!/bin/bash
# (runs on the GPU box through gpurun: rocprofv3 kernel stats of tools/prof_encode.py for each "lib:CLT_GEMM_DEBUG" argument;
# prints the average duration of the GEMM kernels of the fused call)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
run() {  # lib dbg
  export CLT_LIB=$R/clt_training_amd/$1 CLT_GEMM_DEBUG=$2
  rm -rf /tmp/abl_x
  timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl_x -o p -- python3 $R/tools/prof_encode.py > /dev/null 2>&1 || exit 1
  f=$(find /tmp/abl_x -name "*kernel_stats.csv" | head -1)
  echo -n "$1 dbg=$2: "; python3 - $f <<'PY'
import csv,sys
out = []
for r in csv.DictReader(open(sys.argv[1])):
    if ('gemm_' in r['Name'] or 'fused_' in r['Name'] or 'topk_rows' in r['Name']) and int(r['Calls']) == 12:
        out.append(f"{r['Name'][:48]} {float(r['AverageNs'])/1e3:.1f} us")
print("; ".join(out))
PY
}
for a in "$@"; do run ${a%%:*} ${a##*:}; done
