#!/bin/bash
# experiment builds of one encoder kernel file: libcltmi355_abl<name>.so = tuning objects + that file recompiled
# usage: tools/ablate_build.sh [-f file.hip] name1 "flags1" name2 "flags2" ...   (default file: encoder_gemm_w8.hip)
set -e
cd /root/repo/clt_training_amd
F=encoder_gemm_w8
if [ "$1" = "-f" ]; then F=${2%.hip}; shift 2; fi
while [ $# -gt 0 ]; do
  n=$1; f=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DCLT_TUNING $f -c csrc/$F.hip -o build_tuning/abl$n.o 2>/dev/null
  objs=$(ls build_tuning/*.o | grep -v "abl\|$F.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libcltmi355_abl$n.so $objs build_tuning/abl$n.o
done
ls libcltmi355_abl*.so
