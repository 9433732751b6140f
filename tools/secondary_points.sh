#!/bin/bash
# Secondary data points of DESIGN.md section 6 (not the metric): other token counts and the other BASELINE shapes on one
# GPU, 6 timed steps each.  Run through gpurun from the repo root; JSON lines land in gpurun_out/sec_*.json.
set -o pipefail
run() { n=$1; shift; timeout -k 10 280 python bench.py --no-cpu-baseline --steps 6 --warmup 2 "$@" > gpurun_out/sec_$n.json 2> gpurun_out/sec_$n.err || { tail -3 gpurun_out/sec_$n.err; exit 1; }; python tools/show_bench.py gpurun_out/sec_$n.json > gpurun_out/sec_$n.txt; sed -n 1,3p gpurun_out/sec_$n.txt; }
run n1024 --tokens 1024 --steps 20 --warmup 5 &&
run n65536 --tokens 65536 &&
run c3 --sae --layers 1 --d-model 4096 --expansion 32 --k 192 &&
run c2slice --layers 6 --d-model 2048 --expansion 64 --k 64 &&
run concat --coalesce concat &&
run simtp8 --simulate-tp 8
