#!/bin/bash
# sensitivity of the driver-window figure to a loaded host: 16 busy loops beside the bench, with / without polling waits
cd $GRAFT_REPO_ROOT
pids=""
for i in $(seq 16); do python3 -c "
import time
t=time.time()
while time.time()-t < 170: pass" & pids="$pids $!"; done
sleep 1
for mode in default poll default poll; do
  for r in 1 2 3 4 5 6; do
    if [ $mode = poll ]; then export HSA_ENABLE_INTERRUPT=0; else unset HSA_ENABLE_INTERRUPT; fi
    echo -n "$mode: "; DCLR_BENCH_HOSTTRACE=1 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/tmp/err.txt | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), end=' ')"; grep "host trace" /tmp/err.txt | sed 's/.*per step): //' | cut -d' ' -f1,6,16,22-
  done
done
for p in $pids; do kill $p 2>/dev/null; done
wait 2>/dev/null
