#!/bin/bash
cd "$(dirname "$0")/.."
python3 scratch/make_clouds.py kitti 16 16384 /tmp/gauss16.bin
python3 scratch/make_clouds.py ring 16 16384 /tmp/ring16.bin
for f in gauss16 ring16; do
  echo "== table $f"; ./scratch/fps_bench 16384 1024 /tmp/$f.bin 16 | grep -E "^rc|table mode|entries|marks|pairs with" | tail -9
  echo "== wavecand $f"; DCLR_FPS_WAVECAND=1 ./scratch/fps_bench 16384 1024 /tmp/$f.bin 16 | grep -E "^rc|super-rounds" | tail -2
done
python -m pytest tests/test_gpu_ops.py -m gpu -x -q 2>&1 | tail -5
