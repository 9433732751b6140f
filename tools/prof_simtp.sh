#!/bin/bash
# (on the GPU box) rocprofv3 kernel stats of the simulated tp rank: tools/prof_simtp.sh W [bench flags]
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
W=$1; shift
cd /tmp
rm -rf /tmp/ks_sim
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_sim -o p -- python3 $R/bench.py --simulate-tp $W --steps 3 --warmup 1 --no-cpu-baseline "$@" > /tmp/ks_sim.log 2>&1 || { tail -5 /tmp/ks_sim.log; exit 1; }
tail -1 /tmp/ks_sim.log | cut -c1-300
python3 $R/tools/kstats.py /tmp/ks_sim 22
