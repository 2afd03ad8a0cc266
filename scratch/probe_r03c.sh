#!/bin/bash
cd "$(dirname "$0")/.."
python3 scratch/make_clouds.py kitti 16 16384 /tmp/gauss16.bin
python3 scratch/make_clouds.py ring 16 16384 /tmp/ring16.bin
python3 scratch/make_clouds.py ring 8 65536 /tmp/ring64.bin
for f in gauss16 ring16; do
  echo "== sa_bench f16 $f"; ./scratch/sa_bench 1 0 /tmp/$f.bin 16 16384 | tail -1
  echo "== sa_bench f32 $f"; SA_F32=1 ./scratch/sa_bench 1 0 /tmp/$f.bin 16 16384 | tail -1
done
echo "== sa_bench f16 ring64"; ./scratch/sa_bench 1 0 /tmp/ring64.bin 8 65536 | tail -1
echo "== sa_bench f32 ring64"; SA_F32=1 ./scratch/sa_bench 1 0 /tmp/ring64.bin 8 65536 | tail -1
echo "== sa_bench modelnet f16"; ./scratch/sa_bench 1 1 | tail -1
echo "== sa_bench modelnet f32"; SA_F32=1 ./scratch/sa_bench 1 1 | tail -1
