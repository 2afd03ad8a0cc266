#!/bin/bash
# round 5: the driver's 20-step window under a few scheduling variants (each line: pairs/s of three runs)
cd "$(dirname "$0")/.."
run() { python bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.0f' % d['value'], end=' ')"; }
for v in "" "DCLR_BENCH_DENSE_PRIO=0" ; do
  echo -n "env[$v] default: "; for i in 1 2 3; do env $v bash -c "$(declare -f run); run"; done; echo
  echo -n "env[$v] depth 2: "; for i in 1 2 3; do env $v bash -c "$(declare -f run); run --depth 2"; done; echo
  echo -n "env[$v] no-launch-timer: "; for i in 1 2 3; do env $v bash -c "$(declare -f run); run --no-launch-timer"; done; echo
done
