#!/bin/bash
# Two (or N) bench.py processes on one GPU at the same time: does the chip have capacity one process leaves idle?
N=${1:-2}
for i in $(seq 1 $N); do
  python bench.py --no-cpu-baseline --no-launch-timer --steps 300 --warmup 20 > gpurun_out/two_$i.log 2>/dev/null &
done
wait
for i in $(seq 1 $N); do python - <<PY
import json
d=json.loads(open('gpurun_out/two_$i.log').read().strip().splitlines()[-1]); print('proc $i', round(d['value']), d['ms_per_step'])
PY
done
