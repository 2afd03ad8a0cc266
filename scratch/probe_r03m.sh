#!/bin/bash
cd "$(dirname "$0")/.."
python3 scratch/make_clouds.py kitti 16 16384 /tmp/gauss16.bin
python3 scratch/make_clouds.py ring 16 16384 /tmp/ring16.bin
python3 scratch/make_clouds.py ring 8 65536 /tmp/ring64.bin
for f in gauss16 ring16; do
  echo "== sa_bench f16 $f"; ./scratch/sa_bench 1 0 /tmp/$f.bin 16 16384 | tail -2
done
echo "== sa_bench f16 ring64"; ./scratch/sa_bench 1 0 /tmp/ring64.bin 8 65536 | tail -2
echo "== sa_bench modelnet"; ./scratch/sa_bench 1 1 | tail -2
python -m pytest tests/test_gpu_model.py -m gpu -x -q > gpurun_out/r03m_pytest.log 2>&1; echo pytest rc $?; tail -3 gpurun_out/r03m_pytest.log
python3 profiles/collect.py --tag r03m --configs none --modes default,ring,c4,c5,ring_c5 2>&1 | grep -v "^+"
