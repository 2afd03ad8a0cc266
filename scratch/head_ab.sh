#!/bin/bash
# A/B of head-kernel builds: scratch/head_ab.sh name1 name2 ...  (libs scratch/libdeepclr_<name>.so from ab_build.sh);
# per build: the head's time alone at 8 / 80 pairs and the head's own GPU tests on that library.
cd "$(dirname "$0")/.."
echo "== product"; python scratch/head_probe16.py 2>&1 | grep head
for n in "$@"; do
  echo "== $n"; DCLR_LIB=scratch/libdeepclr_$n.so python scratch/head_probe16.py 2>&1 | grep head
  DCLR_LIB=scratch/libdeepclr_$n.so python -m pytest tests/test_gpu_model.py -m gpu -x -q -k "fused_head_chain or matches_golden" 2>&1 | tail -1
done
