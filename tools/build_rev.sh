#!/bin/bash
# libcltmi355_abl<name>.so = the kernels of git revision <rev>, -DCLT_TUNING (same-box A/B against the working tree)
# usage: tools/build_rev.sh <rev> <name>
set -e
rev=$1; name=$2
tmp=/tmp/rev_$name
rm -rf $tmp && mkdir -p $tmp
git -C /root/repo archive $rev clt_training_amd/csrc include | tar -x -C $tmp
cd $tmp/clt_training_amd && mkdir -p b
for f in csrc/*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DCLT_TUNING -c $f -o b/$(basename $f .hip).o 2>/dev/null &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/clt_training_amd/libcltmi355_abl$name.so b/*.o
ls -la /root/repo/clt_training_amd/libcltmi355_abl$name.so
