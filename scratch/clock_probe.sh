#!/bin/bash
# GPU clocks and power while bench.py runs (rocm-smi polled from a second process; read-only).
cd "$(dirname "$0")/.."
( for i in $(seq 1 60); do rocm-smi --showclocks --showpower --json 2>/dev/null | python3 -c "
import sys, json
try:
    d = json.load(sys.stdin); c = d[sorted(d)[0]]
    print({k: v for k, v in c.items() if any(s in k.lower() for s in ('sclk', 'mclk', 'fclk', 'power'))})
except Exception as e:
    print('parse', e)
"; sleep 0.25; done ) > gpurun_out/clock_probe.log 2>&1 &
poll=$!
sleep 1
python bench.py --steps 4000 --warmup 20 --no-cpu-baseline --no-launch-timer 2>/dev/null | cut -c1-100
kill $poll 2>/dev/null
sort gpurun_out/clock_probe.log | uniq -c | sort -rn | head -12
