"""Average rocprofv3 counter values per kernel: python scratch/pmc_summary.py <dir> [name-filter ...]"""
import csv, glob, os, sys
acc = {}
for path in glob.glob(os.path.join(sys.argv[1], '**', '*counter_collection.csv'), recursive=True):
    for row in csv.DictReader(open(path, newline='')):
        k = row['Kernel_Name']
        if len(sys.argv) > 2 and not any(f in k for f in sys.argv[2:]):
            continue
        import re
        m = re.search(r'(\w+kernel\w*(<[^>]*>)?)', k)
        short = m.group(1) if m else k[:40]
        d = acc.setdefault(short, {})
        t, n = d.get(row['Counter_Name'], (0.0, 0))
        d[row['Counter_Name']] = (t + float(row['Counter_Value']), n + 1)
        if 'Start_Timestamp' in row and row.get('End_Timestamp'):
            t, n = d.get('_dur_ns', (0.0, 0))
            d['_dur_ns'] = (t + float(row['End_Timestamp']) - float(row['Start_Timestamp']), n + 1)
for k, d in sorted(acc.items()):
    print(k)
    for c, (t, n) in sorted(d.items()):
        print('   %-32s %16.1f   (n=%d)' % (c, t / n, n))
