#!/bin/bash
# GPU probe of round 3: set abstraction + sampler stamps on Gaussian and ring clouds, host-to-device copy rates
set -x
cd "$(dirname "$0")/.."
python3 scratch/make_clouds.py kitti 16 16384 /tmp/gauss16.bin
python3 scratch/make_clouds.py ring 16 16384 /tmp/ring16.bin
python3 scratch/make_clouds.py ring 8 65536 /tmp/ring64.bin
for f in gauss16 ring16; do
  echo "== sa_bench $f"; ./scratch/sa_bench 1 0 /tmp/$f.bin 16 16384
  echo "== fps_bench $f"; ./scratch/fps_bench 16384 1024 /tmp/$f.bin 16
done
echo "== sa_bench ring64"; ./scratch/sa_bench 1 0 /tmp/ring64.bin 8 65536
python3 scratch/h2d_probe.py
HSA_ENABLE_SDMA=0 python3 scratch/h2d_probe.py
