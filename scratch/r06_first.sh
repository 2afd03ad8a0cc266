#!/bin/bash
# round 6, first GPU call: 32x32-tile flow kernel (product) against the 16x16 form (A/B lib), tests, c4 / c2 lines
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
scratch/flow_ab.sh tile16 > gpurun_out/r06_flow_ab.log 2>&1
cat gpurun_out/r06_flow_ab.log
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r06_gputests.log 2>&1; tail -3 gpurun_out/r06_gputests.log
for lib in "" scratch/libdeepclr_tile16.so; do
  for cfg in c4 c2; do
    echo "== lib=$lib cfg=$cfg"
    DCLR_LIB=${lib:-deepclr_amd/csrc/libdeepclr_amd.so} timeout -k 10 300 python bench.py --config $cfg --steps 200 --warmup 20 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('roofline'))"
  done
done
