#!/bin/bash
cd "$(dirname "$0")/.."
python3 scratch/make_clouds.py kitti 16 16384 /tmp/gauss16.bin
python3 scratch/make_clouds.py ring 16 16384 /tmp/ring16.bin
for j in 4 5 6; do for f in gauss16 ring16; do
  echo "== J=$j $f"; ./scratch/fps_bench_j$j 16384 1024 /tmp/$f.bin 16 | grep -E "^rc|table mode" | tail -2
done; done
python -m pytest tests/test_gpu_ops.py -m gpu -x -q 2>&1 | tail -3
