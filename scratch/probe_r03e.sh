#!/bin/bash
cd "$(dirname "$0")/.."
python3 scratch/make_clouds.py kitti 16 16384 /tmp/gauss16.bin
python3 scratch/make_clouds.py ring 16 16384 /tmp/ring16.bin
python3 scratch/make_clouds.py ring 8 65536 /tmp/ring64.bin
for f in gauss16 ring16; do
  echo "== sa_bench f16 $f"; ./scratch/sa_bench 1 0 /tmp/$f.bin 16 16384 | tail -2
done
echo "== sa_bench f16 ring64"; ./scratch/sa_bench 1 0 /tmp/ring64.bin 8 65536 | tail -2
