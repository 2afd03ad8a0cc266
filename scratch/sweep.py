"""Run bench.py over a grid of (GPU_MAX_HW_QUEUES, depth, extra args) and print one line each (diagnostic)."""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
grid = [a.split(',') for a in sys.argv[1:]]          # e.g. 4,3 8,4 8,6,--ahead,sample
for g in grid:
    env = dict(os.environ, GPU_MAX_HW_QUEUES=g[0])
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--no-cpu-baseline', '--no-launch-timer',
                          '--steps', '60', '--depth', g[1]] + g[2:], env=env, capture_output=True, text=True).stdout
    d = json.loads(out.strip().splitlines()[-1])
    print('queues', g[0], 'depth', g[1], ' '.join(g[2:]), '->', round(d['value']), 'pairs/s', round(d['ms_per_step'], 4), 'ms', flush=True)
