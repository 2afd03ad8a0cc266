#!/bin/bash
cd "$(dirname "$0")/.."
python3 scratch/make_clouds.py kitti 16 16384 /tmp/gauss16.bin
python3 scratch/make_clouds.py ring 16 16384 /tmp/ring16.bin
python3 scratch/make_clouds.py ring 8 65536 /tmp/ring64.bin
for f in gauss16 ring16; do
  echo "== table $f"; ./scratch/fps_bench 16384 1024 /tmp/$f.bin 16 | grep -E "^rc|table mode" | tail -2
  echo "== wavecand $f"; DCLR_FPS_WAVECAND=1 ./scratch/fps_bench 16384 1024 /tmp/$f.bin 16 | grep -E "^rc|super-rounds" | tail -2
  echo "== sa_bench f16 $f"; ./scratch/sa_bench 1 0 /tmp/$f.bin 16 16384 | tail -2
done
echo "== sa_bench f16 ring64"; ./scratch/sa_bench 1 0 /tmp/ring64.bin 8 65536 | tail -2
python -m pytest tests/test_gpu_ops.py -m gpu -x -q 2>&1 | tail -3
