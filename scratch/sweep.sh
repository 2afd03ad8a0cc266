#!/bin/bash
# round 5: pipeline shape sweep (side streams x batches per launch), 200 steps each
cd "$(dirname "$0")/.."
for cfg in "--depth 2" "--depth 3" "--depth 4" "--depth 3 --group 5 --steps 200" "--depth 2 --group 20 --steps 200" "--depth 3 --group 20 --steps 200"; do
  python bench.py $cfg --no-secondary --no-cpu-baseline --no-launch-timer 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$cfg', round(d['value']), d['ms_per_step'])"
done
