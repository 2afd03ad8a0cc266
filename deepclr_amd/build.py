"""Build libdeepclr_amd.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

Every object is compiled with -Rpass-analysis=kernel-resource-usage; the remarks are condensed into
csrc/<name>.usage.txt (kernel, VGPRs, scratch bytes, occupancy, LDS bytes). kernel_usage() reads them:
tests/test_host.py keeps the hot kernels free of scratch (a spill in the sampler cost 20 % once)."""
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc')
LIB = os.path.join(CSRC, 'libdeepclr_amd.so')
SOURCES = ['api.hip', 'fps.hip', 'grouping.hip', 'knn.hip', 'sa.hip', 'gemm.hip', 'flow.hip', 'gemm16.hip', 'flow16.hip', 'prep.hip', 'forward.hip']
HEADERS = ['common.h', 'mma.h', 'mma16f.h', os.path.join('..', '..', 'include', 'deepclr_amd.h')]
# -ffp-contract=off: the distance recipe shared with the oracle is one rounding per operation;
# MLP code requests FMA explicitly.
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-Wall',
         '-Wno-unused-function']


def _hipcc() -> str:
    for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if cand and (os.path.sep not in cand or os.path.exists(cand)):
            return cand
    raise RuntimeError('hipcc not found')


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


_REMARK = re.compile(r'remark: (?:Function Name: (?P<name>\S+)|\s+(?P<key>[A-Za-z ]+?)(?: \[[^\]]*\])?: (?P<val>\d+))')
_KEYS = {'VGPRs': 'vgprs', 'AGPRs': 'agprs', 'ScratchSize': 'scratch', 'Occupancy': 'occupancy', 'LDS Size': 'lds',
         'SGPRs': 'sgprs'}


def _write_usage(path: str, stderr: str) -> str:
    """Condense the resource-usage remarks of one compile into `path`; returns the other diagnostics."""
    rows, cur, rest = [], None, []
    for line in stderr.splitlines():
        if 'remark:' not in line or 'kernel-resource-usage' not in line:
            if cur is None or line.strip() not in ('', '^'):
                rest.append(line)
            continue
        m = _REMARK.search(line)
        if not m:
            continue
        if m.group('name'):
            cur = {'kernel': m.group('name')}
            rows.append(cur)
        elif cur is not None and m.group('key') in _KEYS:
            cur[_KEYS[m.group('key')]] = int(m.group('val'))
    with open(path, 'w') as f:
        for r in rows:
            f.write('{kernel} vgprs={vgprs} agprs={agprs} scratch={scratch} occupancy={occupancy} lds={lds}\n'.format(
                **{k: r.get(k, 0) for k in ('kernel', 'vgprs', 'agprs', 'scratch', 'occupancy', 'lds')}))
    return '\n'.join(l for l in rest if 'kernel-resource-usage' not in l and not l.startswith(' ') or 'error' in l or 'warning' in l)


def kernel_usage() -> dict:
    """{mangled kernel name: {'vgprs', 'agprs', 'scratch', 'occupancy', 'lds'}} from the last build's usage files."""
    out = {}
    for src in SOURCES:
        path = os.path.join(CSRC, src.replace('.hip', '.usage.txt'))
        if not os.path.exists(path):
            continue
        for line in open(path):
            name, *fields = line.split()
            out[name] = {k: int(v) for k, v in (f.split('=') for f in fields)}
    return out


def build(force: bool = False, verbose: bool = False) -> str:
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, h) for h in HEADERS]
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace('.hip', '.o'))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            jobs.append([hipcc, *FLAGS, '-c', s, '-o', o])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    def compile_one(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        res = subprocess.run(cmd + ['-Rpass-analysis=kernel-resource-usage'], stderr=subprocess.PIPE, text=True)
        rest = _write_usage(cmd[-1].replace('.o', '.usage.txt'), res.stderr)
        if rest.strip():
            sys.stderr.write(rest)
        if res.returncode != 0:
            raise subprocess.CalledProcessError(res.returncode, cmd)

    with ThreadPoolExecutor(max_workers=4) as pool:
        list(pool.map(compile_one, jobs))
    if force or jobs or _stale(LIB, objs):
        run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB, *objs])
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
