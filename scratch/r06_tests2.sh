#!/bin/bash
cd "$(dirname "$0")/.."
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r06_gputests.log 2>&1; tail -4 gpurun_out/r06_gputests.log
for i in 1 2; do timeout -k 10 200 python bench.py --latency --steps 50 --warmup 10 --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('latency', d['ms_per_step'], d['latency_ms'])"; done
