#!/bin/bash
# experiment builds of the 16x16x32 encoder kernel: libcltmi355_abl<name>.so = tuning objects + that file recompiled
# usage: tools/ablate_build.sh name1 "flags1" name2 "flags2" ...
set -e
cd /root/repo/clt_training_amd
while [ $# -gt 0 ]; do
  n=$1; f=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DCLT_TUNING $f -c csrc/encoder_gemm_fast16.hip -o build_tuning/abl$n.o 2>/dev/null
  objs=$(ls build_tuning/*.o | grep -v "abl\|encoder_gemm_fast16.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libcltmi355_abl$n.so $objs build_tuning/abl$n.o
done
ls libcltmi355_abl*.so
