#!/bin/bash
cd "$(dirname "$0")/.."
for rep in 1 2; do
for lib in deepclr_amd/csrc/libdeepclr_amd.so scratch/libdeepclr_sanoovf.so; do
  for cfg in c4 c2; do
    echo "== $lib $cfg"
    DCLR_LIB=$lib timeout -k 10 200 python bench.py --config $cfg --alone-only 6 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:v for k,v in d['kernels_alone_us'].items() if 'sa_msg' in k or 'flow' in k})"
  done
done
done
