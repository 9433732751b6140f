#!/bin/bash
# (on the GPU box) the benchmark step with two or more libraries, one process each, same box, twice: tools/ab_step.sh lib1 lib2 ...
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
for l in "$@"; do
  echo "== $l"
  CLT_LIB=$R/clt_training_amd/$l timeout -k 10 200 python3 -u $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline > /tmp/ab_step.json 2>/dev/null || exit 1
  python3 $R/tools/show_bench.py /tmp/ab_step.json | grep -v "^$"
done
done
