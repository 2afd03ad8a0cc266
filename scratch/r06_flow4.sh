#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
python -m pytest tests/test_gpu_model.py -m gpu -x -q -k "flow_embedding_split or matches_golden or radius_mask or unfilled" 2>&1 | tail -3
echo "== product"; python scratch/flow_probe.py 2>&1 | grep pairs
for n in "$@"; do
  echo "== $n"; DCLR_LIB=scratch/libdeepclr_$n.so python scratch/flow_probe.py 2>&1 | grep pairs
done
DCLR_LIB=scratch/libdeepclr_st32.so python scratch/flow_stamps.py 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r06_flow4.log 2>&1
cat gpurun_out/r06_flow4.log
