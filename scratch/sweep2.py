import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for spec in sys.argv[1:]:
    envs, *args = spec.split(',')
    env = dict(os.environ)
    for kv in envs.split(';'):
        if kv:
            k, v = kv.split('='); env[k] = v
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--no-cpu-baseline', '--no-launch-timer',
                          '--steps', '60'] + args, env=env, capture_output=True, text=True).stdout
    d = json.loads(out.strip().splitlines()[-1])
    print(spec, '->', round(d['value']), 'pairs/s', round(d['ms_per_step'], 4), 'ms', flush=True)
