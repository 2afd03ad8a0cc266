"""bench.py -> short summary: throughput and per-kernel live / alone times (diagnostic)."""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--no-cpu-baseline'] + sys.argv[1:],
                     capture_output=True, text=True)
try:
    d = json.loads(out.stdout.strip().splitlines()[-1])
except Exception:
    print(out.stdout[-2000:], out.stderr[-3000:]); raise SystemExit(1)
print(round(d['value']), 'pairs/s', round(d['ms_per_step'], 4), 'ms/step')
for r in d.get('rooflines', []):
    print('  {:18s} live {:7.1f} us  alone {:7.1f} us  frac {:.3f}  frac_alone {:.3f}'.format(
        r['kernel'], r['avg_us'], r['alone_us'] or 0, r['frac'], r['frac_alone'] or 0))
print('  all:', d.get('kernels_us'))
