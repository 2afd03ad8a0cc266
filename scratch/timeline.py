"""Timeline of the kernels of a short bench window from a rocprofv3 kernel trace (csv)."""
import csv, glob, sys
rows = []
for path in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(path)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Stream_Id', r.get('Queue_Id', '?'))))
rows.sort()
keep = [r for r in rows if any(k in r[2] for k in ('fps_', 'sa_msg', 'flow16', 'head16', 'knn_rows', 'linear_kernel', 'fc_kernel'))]
# the timed window = the last 2 head16 launches; print everything from 1 ms before the first of them
heads = [r for r in keep if 'head16' in r[2]]
t0 = heads[-2][0] - 2_000_000 if len(heads) >= 2 else keep[0][0]
tend = max(r[1] for r in keep)
print('window %.3f ms' % ((tend - t0) / 1e6))
for s, e, name, q in keep:
    if e < t0:
        continue
    short = name.split('::')[-1].split('(')[0][:28]
    print('%8.3f -> %8.3f ms  (%7.1f us)  q%s  %s' % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e3, q, short))
