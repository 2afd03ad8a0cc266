#!/bin/bash
# A/B of flow-embedding builds: scratch/flow_ab.sh name1 name2 ...  (libs scratch/libdeepclr_<name>.so from ab_build.sh);
# per build: the kernel's time alone (k = 20 at 8 / 80 pairs, k = 30 at 256 pairs) and the flow-embedding GPU tests on that library.
cd "$(dirname "$0")/.."
echo "== product"; python scratch/flow_probe.py 2>&1 | grep pairs
for n in "$@"; do
  echo "== $n"; DCLR_LIB=scratch/libdeepclr_$n.so python scratch/flow_probe.py 2>&1 | grep pairs
  DCLR_LIB=scratch/libdeepclr_$n.so python -m pytest tests/test_gpu_model.py -m gpu -x -q -k "flow_embedding_split or matches_golden or radius_mask or unfilled" 2>&1 | tail -1
done
