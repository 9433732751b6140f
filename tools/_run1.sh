set -e
cd /root/repo
for r in 1 2; do
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 12 --warmup 4 > gpurun_out/ov0.json 2>gpurun_out/ov0.err
python tools/show_bench.py gpurun_out/ov0.json | sed -n "1p"
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 12 --warmup 4 --routes-on overlap_single > gpurun_out/ov1.json 2>gpurun_out/ov1.err
python tools/show_bench.py gpurun_out/ov1.json | sed -n "1p"
done
python tools/show_bench.py gpurun_out/ov1.json
