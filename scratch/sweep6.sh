#!/bin/bash
# round 6: pipeline shape sweep around the shipped 3 x 10 (side streams x batches per sampling / dense launch), 240 steps each
cd "$(dirname "$0")/.."
for cfg in "--depth 3 --group 10" "--depth 3 --group 8" "--depth 3 --group 12" "--depth 2 --group 16" "--depth 3 --group 16" "--depth 4 --group 8" "--depth 2 --group 12" "--depth 3 --group 10"; do
  timeout -k 10 200 python bench.py $cfg --steps 240 --warmup 48 --no-secondary --no-cpu-baseline --no-launch-timer 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$cfg', round(d['value']), d['ms_per_step'])"
done
