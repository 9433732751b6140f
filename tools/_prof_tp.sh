#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp; rm -rf /tmp/ptp
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ptp -o p -- python3 $R/bench.py --no-cpu-baseline --steps 4 --warmup 2 --simulate-tp 8 > /dev/null 2>&1 || exit 1
cp $(find /tmp/ptp -name "*kernel_stats.csv" | head -1) $R/gpurun_out/tp8_kernel_stats.csv
