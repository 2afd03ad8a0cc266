#!/bin/bash
# round 6: the judged evidence -- kernel stats (contended + alone), PMC traffic / mfma_busy, then the driver's own command and the
# lines beside it (profiles/collect.py); rocprofv3 wants a writable cwd-independent TMPDIR
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
python3 profiles/collect.py --tag r06 --configs c2,c4,c5 --modes driver,default,serial,strict,ring,h2d,latency,c4,c5,ring_c5 > gpurun_out/r06_collect.log 2>&1
echo "collect rc=$?"
tail -30 gpurun_out/r06_collect.log
