#!/bin/bash
# round 6: flow kernel variants alone (flow_probe) + the flow tests on the candidates
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
echo "== product (32x32 tiles, read-ahead at RT 4)"; python scratch/flow_probe.py 2>&1 | grep pairs
for n in tile16 t16nm f32a1 f32a2 f32a3; do
  echo "== $n"; DCLR_LIB=scratch/libdeepclr_$n.so python scratch/flow_probe.py 2>&1 | grep pairs
done
for n in t16nm; do
  DCLR_LIB=scratch/libdeepclr_$n.so python -m pytest tests/test_gpu_model.py -m gpu -x -q -k "flow_embedding_split or matches_golden or radius_mask or unfilled" 2>&1 | tail -1
done
python -m pytest tests/test_gpu_model.py -m gpu -x -q -k "flow_embedding_split or matches_golden or radius_mask or unfilled" 2>&1 | tail -1
} > gpurun_out/r06_flow2.log 2>&1
cat gpurun_out/r06_flow2.log
