#!/bin/bash
# --strict (one batch of 8 pairs per sampling launch and per dense launch) at 2..6 side streams, and a kernel timeline of
# depth 3 and depth 4 (rocprofv3 --kernel-trace): what bounds the un-fused regime and why one more stream did not help.
cd "$(dirname "$0")/.."
for d in 2 3 4 5 6 8; do
  echo -n "strict depth $d: "; python bench.py --strict --depth $d --steps 200 --warmup 20 --no-cpu-baseline --no-launch-timer 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), 'pairs/s', round(1e3*d['ms_per_step']), 'us/step')"
done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for d in 3 4; do
  rm -rf gpurun_out/strict_trace_$d
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/strict_trace_$d -- python3 bench.py --strict --depth $d --steps 40 --warmup 20 --no-cpu-baseline --no-launch-timer > /dev/null 2>&1
  python3 - gpurun_out/strict_trace_$d $d <<'PY'
import csv, glob, sys
rows = []
for path in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(path)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
def kind(n):
    for k, s in (('fps_', 'fps'), ('sa_msg', 'sa'), ('flow16', 'flow'), ('head16', 'head'), ('knn_rows', 'knn'), ('linear_kernel', 'lin'), ('fc_kernel', 'fc')):
        if k in n: return s
    return None
keep = [(s, e, kind(n)) for s, e, n in rows if kind(n)]
fps = [r for r in keep if r[2] == 'fps']
# steady state: the last 30 sampler launches
fps = fps[-30:]
t0, t1 = fps[0][0], fps[-1][0]
gap = (t1 - t0) / (len(fps) - 1) / 1e3
dur = sum(e - s for s, e, _ in fps) / len(fps) / 1e3
# how many sampler launches overlap in time, on average
events = sorted([(s, 1) for s, e, _ in fps] + [(e, -1) for s, e, _ in fps])
cur, last, acc = 0, events[0][0], 0
for t, dv in events:
    acc += cur * (t - last); last = t; cur += dv
conc = acc / (events[-1][0] - events[0][0])
busy = {}
for s, e, k in keep:
    if s >= t0 and e <= fps[-1][1]:
        busy[k] = busy.get(k, 0) + (e - s)
span = fps[-1][1] - t0
print('depth %s: sampler launch every %.0f us, lasting %.0f us, %.2f in flight on average; kernel time / wall time per kind: %s'
      % (sys.argv[2], gap, dur, conc, {k: round(v / span, 2) for k, v in busy.items()}))
PY
  rm -rf gpurun_out/strict_trace_$d
done
