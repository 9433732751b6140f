#!/bin/bash
# (on the GPU box) row passes, one process per library, same box: tools/ab_rows.sh lib1 lib2 ... [-- bench_rows flags]
R=$GRAFT_REPO_ROOT
libs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do libs+=("$1"); shift; done; shift
for rep in 1 2; do
for l in "${libs[@]}"; do
  echo "== $l"; CLT_LIB=$R/clt_training_amd/$l timeout -k 10 120 python3 -u $R/tools/bench_rows.py "$@" 2>/dev/null | grep "us"
done
done
