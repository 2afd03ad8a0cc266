#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
d=${1:-4}
rm -rf gpurun_out/strict_trace_$d
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/strict_trace_$d -- python3 bench.py --strict --depth $d --steps 40 --warmup 20 --no-cpu-baseline --no-launch-timer > /dev/null 2>&1
python3 - gpurun_out/strict_trace_$d <<'PY'
import csv, glob, sys
rows = []
for path in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    rd = csv.DictReader(open(path))
    for r in rd:
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '?'), r.get('Stream_Id', '?')))
rows.sort()
def kind(n):
    for k, s in (('fps_', 'fps'), ('sa_msg', 'sa'), ('flow16', 'flow'), ('head16', 'head'), ('knn_rows', 'knn'), ('linear_kernel', 'lin'), ('fc_kernel', 'fc')):
        if k in n: return s
    return None
keep = [(s, e, kind(n) or n[:30], q, st) for s, e, n, q, st in rows]
fps = [r for r in keep if r[2] == 'fps']
t0 = fps[-8][0]
for s, e, k, q, st in keep:
    if s >= t0:
        print('%8.3f -> %8.3f ms (%6.1f us) queue %s stream %s %s' % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e3, q, st, k))
PY
rm -rf gpurun_out/strict_trace_$d
