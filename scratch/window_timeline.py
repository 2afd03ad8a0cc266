"""Kernel timeline of the driver's 20-step window from a rocprofv3 kernel trace: every kernel of the last two dense launches'
window with its stream/queue, plus the chip-time accounting (sum of duration x min(1, workgroups / 256))."""
import csv, glob, sys
rows = []
for path in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(path)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '?'),
                     int(r.get('Grid_Size', r.get('Grid_Size_X', 0)) or 0), int(r.get('Workgroup_Size', r.get('Workgroup_Size_X', 1)) or 1)))
rows.sort()
def kind(n):
    for k, s in (('fps_', 'fps'), ('sa_msg', 'sa'), ('flow16', 'flow'), ('head16', 'head'), ('knn_rows', 'knn'), ('linear_kernel', 'lin'), ('fc_kernel', 'fc')):
        if k in n: return s
    return None
keep = [(s, e, kind(n), q, g, w) for s, e, n, q, g, w in rows if kind(n)]
# bursts of kernels separated by > 0.15 ms of nothing (fences, host work); the timed window is the burst that holds exactly
# two sampler launches and two head launches of the grouped sizes (argv[2]: which such burst, default the first)
bursts, cur, last_end = [], [], None
for r in keep:
    if last_end is not None and r[0] - last_end > 150_000:
        bursts.append(cur); cur = []
    cur.append(r); last_end = max(last_end or 0, r[1])
bursts.append(cur)
cands = [b for b in bursts if sum(1 for r in b if r[2] == 'fps') == 2 and sum(1 for r in b if r[2] == 'head') == 2]
print('bursts:', [(len(b), sum(1 for r in b if r[2] == 'fps'), sum(1 for r in b if r[2] == 'head')) for b in bursts])
win = cands[int(sys.argv[2]) if len(sys.argv) > 2 else 0]
t0 = min(r[0] for r in win); tend = max(r[1] for r in win)
print('window %.3f ms, %d kernels' % ((tend - t0) / 1e6, len(win)))
busy = 0.0
for s, e, k, q, g, w in win:
    wgs = g // max(w, 1) if g else 0
    share = min(1.0, wgs / 256.0) if wgs else 1.0
    busy += (e - s) * share
    print('%8.3f -> %8.3f ms (%7.1f us) q%-3s %-5s %6d workgroups' % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e3, q, k, wgs))
print('sum of duration x min(1, workgroups / 256): %.3f ms (an upper bound of the chip-time: overlapping kernels share CUs)' % (busy / 1e6))
