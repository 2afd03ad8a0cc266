#!/bin/bash
# round 6: the GPU suite + the driver's own command A/B (eager dense enqueue on / off)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r06_gputests.log 2>&1; tail -5 gpurun_out/r06_gputests.log
grep -q "passed" gpurun_out/r06_gputests.log && ! grep -q "failed" gpurun_out/r06_gputests.log || exit 1
for rep in 1 2 3; do
  for e in 1 0; do
    echo "== driver window, eager=$e (rep $rep)"
    timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --eager-dense $e 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), d['ms_per_step'])"
  done
done
for e in 1 0; do
  echo "== 200 steps, eager=$e"
  timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --eager-dense $e 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), d['ms_per_step'])"
done
