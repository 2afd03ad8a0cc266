#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
echo "== product"; python scratch/flow_probe.py 2>&1 | grep pairs
for n in "$@"; do
  echo "== $n"; DCLR_LIB=scratch/libdeepclr_$n.so python scratch/flow_probe.py 2>&1 | grep pairs
done
} > gpurun_out/r06_flow3.log 2>&1
cat gpurun_out/r06_flow3.log
