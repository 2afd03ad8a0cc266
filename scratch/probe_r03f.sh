#!/bin/bash
cd "$(dirname "$0")/.."
python3 scratch/make_clouds.py kitti 16 16384 /tmp/gauss16.bin
./scratch/fps_bench 16384 1024 /tmp/gauss16.bin 16 | grep -E "^rc|table mode|pairs with" | tail -7
python -m pytest tests -m gpu -x -q > gpurun_out/r03f_pytest.log 2>&1; echo pytest rc $?; tail -3 gpurun_out/r03f_pytest.log
python3 profiles/collect.py --tag r03f --configs none --modes driver,default,strict,ring,latency 2>&1 | grep -v "^+"
