#!/usr/bin/env python3
"""Throughput of the DeepCLR forward hot path on MI355X: scan-pairs/s (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W [--config c2|c4|c5]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

One step = one forward pass of the whole hot path (FPS -> set abstraction -> kNN -> flow embedding ->
pose head) over one batch of synthetic pairs already resident in HBM. The headline workload (`--config c2`,
the default) is BASELINE.json configs[1]: 8 KITTI-sized pairs of 2 x 16384 points per GPU; `c4` (256 ModelNet
pairs of 2 x 2048 points) and `c5` (4 pairs of 2 x 65536 points) are BASELINE.json configs[3] / configs[4] and
print the same JSON schema. Ranks own independent pairs (weak scaling, no data-path collective); the only
exchange is an RCCL all-gather of the pose outputs.

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (one fresh child process
per GPU, before this process has touched the GPU) and relays rank 0's JSON line. Rank 0 prints ONE JSON line.
The CPU oracle is used only for the `cpu_baseline` leg and the pose check -- never inside the timed region.

The plain single-GPU command (`python bench.py [--gpus 1 --steps K --warmup W]`, the driver's) measures the headline window first
(a child process of its own) and then attaches a `secondary` block: steady state over 200 steps, --strict, ring scans, c4, c5 and
the single-pair latency, each measured by a further child (pairs/s, ms/step, pose delta max, dominant kernel's roofline fraction);
`--no-secondary` prints the headline alone. This parent process never touches the GPU.

Lines beside the headline (same schema, `config.mode` says which; profiles/r05_*):
  --strict    one batch of B pairs per sampling launch and per dense launch (no cross-batch fusion)
  --h2d       every step's batch is copied from pinned host memory inside the loop (copy stream, overlapped),
              as the reference does per pair (/root/reference/scripts/inference.py:89-90)
  --clouds ring   LiDAR-density clouds (64 rings x azimuth, deepclr_amd/synthetic.py:ring_scan): ball queries reach
              their nsample caps, which the Gaussian clouds of the headline never do
  --latency   the reference's own call pattern: one pair per ModelInferenceHelper.predict call, per-pair ms between
              events as /root/reference/scripts/timing.py:32-47 takes it (pairwise and sequential)
"""
import argparse
import hashlib
import json
import os
import signal
import socket
import subprocess
import sys
import time

# The HIP runtime multiplexes all streams of a process onto 4 hardware queues by default. The runner uses the
# main stream + 3 side streams, and RCCL brings its own stream for the all-gather: with 4 queues that stream
# shares a queue with a ~1 ms sampling kernel and every step waits for it (measured on one GPU with a
# one-rank process group: 18.6k instead of 28.9k pairs/s). Must be set before the runtime initialises.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

import numpy as np                                          # noqa: E402
import torch                                                # noqa: E402  (importing torch does not touch the GPU)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from deepclr_amd import ops, synthetic                      # noqa: E402
from deepclr_amd.config import model_config_from_dict       # noqa: E402
from deepclr_amd.labels import LabelType                    # noqa: E402
from deepclr_amd.models import build_model                  # noqa: E402
from deepclr_amd.pipeline import HostBatchFeeder, PipelinedForward, PipelinedSequence           # noqa: E402

# BASELINE.json configs that fit one GPU. depth / group: side streams and batches per sampling launch of the
# pipelined runner (the sampler is one workgroup per cloud: c2 needs grouped launches to have enough clouds in
# flight, c4 already brings 512 clouds per batch).
CONFIGS = {
    'c2': {'kind': 'kitti', 'pairs': 8, 'points': 16384, 'depth': 3, 'group': 10, 'dense_group': 1, 'steps': 200, 'warmup': 20,
           'baseline': 'BASELINE.json configs[1]'},
    'c4': {'kind': 'modelnet', 'pairs': 256, 'points': 2048, 'depth': 2, 'group': 1, 'dense_group': 0, 'steps': 40, 'warmup': 5,
           'baseline': 'BASELINE.json configs[3]'},
    'c5': {'kind': 'kitti', 'pairs': 4, 'points': 65536, 'depth': 2, 'group': 20, 'dense_group': 1, 'steps': 200, 'warmup': 40,
           'baseline': 'BASELINE.json configs[4]'},
}
PAIRS_PER_GPU = CONFIGS['c2']['pairs']
POINTS = CONFIGS['c2']['points']
FP32_MATRIX_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, = fp32 vector peak
FP32_VECTOR_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: peak FP32 (vector)
F16_MATRIX_PEAK_TFLOPS = 2516.6      # MI355X_MICROARCH.md: dense f16/bf16 MFMA = 16 x the f32 matrix rate (~2.5 PF)
SPLIT_PRODUCTS = 3                   # f16 MFMAs per f32-accurate product on the split path (csrc/mma16f.h)
HBM_PEAK_GBS = 8000.0                # MI355X_MICROARCH.md: HBM3E spec peak
FPS_FLOP_PER_EVAL = 9                # 3 sub, 3 mul, 2 add, 1 min per (point, sample) distance update
TRAFFIC_FILE = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')


def kernel_source_hash() -> str:
    """sha256 over the kernel sources: PMC traffic figures collected offline are attached to a bench line only
    when they were measured on exactly these kernels (profiles/collect_traffic.py stamps the same hash)."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, 'deepclr_amd', 'csrc')
    for name in sorted(os.listdir(csrc)):
        if name.endswith(('.hip', '.h')):
            with open(os.path.join(csrc, name), 'rb') as fh:
                h.update(name.encode() + b'\0' + fh.read())
    return h.hexdigest()[:16]


# ----------------------------------------------------------------------------------------------------------
# self-launch: `python bench.py --gpus N` (what the driver runs) -> N fresh children, one per GPU
# ----------------------------------------------------------------------------------------------------------
def rank_environments(n: int, port: int, base_env=None):
    """The N child environments torch.distributed.run would build for one node."""
    envs = []
    for rank in range(n):
        env = dict(os.environ if base_env is None else base_env)
        env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), DCLR_BENCH_CHILD='1')
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        env.setdefault('GPU_MAX_HW_QUEUES', '8')
        envs.append(env)
    return envs


def rank_cores(local_rank: int, local_world: int, available=None):
    """The host cores of one rank: the cores this process may run on, dealt in contiguous, disjoint slices to the ranks of
    the node (8 ranks share one host: eight Python host loops, their HIP runtime threads and RCCL's proxy threads otherwise
    migrate over -- and contend for -- the same cores). Fewer cores than ranks: no pinning (None)."""
    cores = sorted(os.sched_getaffinity(0) if available is None else available)
    if local_world <= 1 or len(cores) < local_world:
        return None
    per = len(cores) // local_world
    return cores[local_rank * per:(local_rank + 1) * per]


def pin_rank(local_rank: int, local_world: int):
    """Pin this process (and the threads it starts later) to its slice; DCLR_BENCH_PIN=0 leaves the affinity alone."""
    if os.environ.get('DCLR_BENCH_PIN', '1') == '0':
        return None
    cores = rank_cores(local_rank, local_world)
    if cores:
        os.sched_setaffinity(0, cores)
        torch.set_num_threads(max(1, min(len(cores), torch.get_num_threads())))
    return cores


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def spawn_ranks(n: int, argv) -> int:
    """Start one child per GPU BEFORE this process makes any GPU call (a process that has initialised the GPU
    must never exec or fork GPU children). Rank 0's stdout is relayed; returns the worst exit code."""
    envs = rank_environments(n, free_port())
    cmd = [sys.executable, os.path.abspath(__file__)] + list(argv)
    procs = []
    for rank, env in enumerate(envs):
        out = subprocess.PIPE if rank == 0 else subprocess.DEVNULL
        procs.append(subprocess.Popen(cmd, env=env, stdout=out, text=(rank == 0)))
    line_seen = False
    for line in procs[0].stdout:
        sys.stdout.write(line)
        sys.stdout.flush()
        line_seen = line_seen or line.lstrip().startswith('{')
    codes = []
    deadline = time.time() + 120
    for p in procs:
        try:
            codes.append(p.wait(timeout=max(1.0, deadline - time.time())))
        except subprocess.TimeoutExpired:
            p.kill()                                        # the exact child started here, nothing else
            codes.append(p.wait())
    bad = [c for c in codes if c != 0]
    if bad or not line_seen:
        sys.stderr.write('bench.py: rank exit codes {} (json line seen: {})\n'.format(codes, line_seen))
        return bad[0] if bad else 1
    return 0


# ----------------------------------------------------------------------------------------------------------
# per-kernel HIP events
# ----------------------------------------------------------------------------------------------------------
class LaunchTimer:
    """HIP events around library launches, on the stream the kernel is enqueued on. Every SAMPLE_EVERY-th launch
    of each kind is bracketed; counted per kind, not per step, so that launches covering several batches (one
    sampling / dense launch per `group` steps) are sampled at the same rate whatever the warm-up count. With ten
    batches per launch there are < 1 spans per step and every launch is bracketed (the driver's 20-step window
    holds two dense launches); when every batch had launches of its own the events cost ~10 % and every third
    launch was sampled."""

    SAMPLE_EVERY = 1

    POOL = 256                       # events created up front: creation is host time inside a short timed window

    RAW_POOL = 64                    # dclr_merge_forward / dclr_cloud_forward calls bracketed before the pool has to grow inside the window

    def __init__(self, sample_every=None, raw_pool=None):
        raw_pool = self.RAW_POOL if raw_pool is None else raw_pool
        self.spans = []
        self.raw = []
        self.calls = {}
        self._hip = None
        from deepclr_amd import lib as _lib
        self._stream_ptr = _lib.stream_ptr
        self.main_stream = torch.cuda.current_stream().cuda_stream
        if sample_every is not None:
            self.SAMPLE_EVERY = sample_every
        self._pool = []
        self._raw_pool = []          # arrays of MERGE_EVENTS raw hipEvent_t, created here, destroyed by close()
        self._raw_all = []
        try:
            self._pool = [torch.cuda.Event(enable_timing=True) for _ in range(self.POOL)]
        except Exception:            # CPU-only unit tests patch torch.cuda.Event; a missing device must not matter here
            self._pool = []
        if raw_pool and self._pool:
            try:
                self._hip = self._load_hip()
            except OSError:
                raw_pool = 0
            for _ in range(raw_pool):
                arr = self._new_raw()
                if arr is None:
                    break
                self._raw_pool.append(arr)

    @staticmethod
    def _load_hip():
        import ctypes
        return ctypes.CDLL('libamdhip64.so')

    def _new_raw(self):
        import ctypes
        from deepclr_amd import lib
        arr = (ctypes.c_void_p * lib.MERGE_EVENTS)()
        for i in range(lib.MERGE_EVENTS):
            ev = ctypes.c_void_p()
            if self._hip.hipEventCreate(ctypes.byref(ev)) != 0:
                return None
            arr[i] = ev.value
        self._raw_all.append(arr)
        return arr

    def close(self):
        """Destroy the raw events (after summary(): a 200-step run or a solo pass otherwise leaks ten handles per call)."""
        import ctypes
        for arr in self._raw_all:
            for h in arr:
                if h:
                    self._hip.hipEventDestroy(ctypes.c_void_p(h))
        self._raw_all, self._raw_pool, self.raw = [], [], []

    def _event(self):
        return self._pool.pop() if self._pool else torch.cuda.Event(enable_timing=True)

    def next_step(self):
        pass

    def _sampled(self, kind: str) -> bool:
        n = self.calls.get(kind, 0)
        self.calls[kind] = n + 1
        return n % self.SAMPLE_EVERY == 0

    def begin(self, name):
        if not self._sampled(name):
            return None
        start = self._event()
        start.record()
        return (name, start, self._stream_ptr() == self.main_stream)

    def end(self, token):
        if token is None:
            return
        stop = self._event()
        stop.record()
        self.spans.append((token[0], token[1], stop, token[2]))

    def merge_events(self, pairs, npoint, k, n_fc=3, stages=3):
        """Raw HIP events for the stages of one dclr_merge_forward call (the dense stages are one foreign call;
        the library records these between its launches, on the launch stream). stages: the call's stage mask
        (1 = layer-1 halves + kNN, 2 = flow embedding + head + FC tail)."""
        rows = pairs * npoint
        names = ['linear_pair[2x%dx128x64]' % rows, None, 'knn_rows[%dx%dx%d]' % (pairs, npoint, k),
                 'flow_embedding[%dx%dx%d]' % (pairs, npoint, k), 'head_conv_fused[%dx%d]' % (pairs, npoint)]
        names += ['fc[%d]' % pairs] * n_fc
        live = [n for i, n in enumerate(names) if n is not None and ((stages & 1 and i < 3) or (stages & 2 and i >= 3))]
        for n in set(live):
            self.calls[n] = self.calls.get(n, 0) + live.count(n)
        if not self._sampled('merge_forward/%d' % stages):
            return None
        if self._hip is None:
            self._hip = self._load_hip()
        arr = self._raw_pool.pop() if self._raw_pool else self._new_raw()
        if arr is None:
            return None
        on_main = self._stream_ptr() == self.main_stream
        self.raw.append((arr, names, on_main))
        return arr

    def cloud_events(self, clouds, n):
        """Raw HIP events for the two per-cloud stages of one dclr_cloud_forward call (start, after sampling, after set
        abstraction), recorded by the library on the launch stream; the stage-1 events of the same call come from
        merge_events(..., stages=1)."""
        names = ['fps_clouds[%dx%d]' % (clouds, n), 'sa_msg_fused[%dx%d]' % (clouds, n)]
        for nm in names:
            self.calls[nm] = self.calls.get(nm, 0) + 1
        if not self._sampled('cloud_forward'):
            return None
        if self._hip is None:
            self._hip = self._load_hip()
        arr = self._raw_pool.pop() if self._raw_pool else self._new_raw()
        if arr is None:
            return None
        self.raw.append((arr, names, self._stream_ptr() == self.main_stream))
        return arr

    def summary(self):
        acc = {}
        for name, a, b, on_main in self.spans:
            tot, cnt, _ = acc.get(name, (0.0, 0, on_main))
            acc[name] = (tot + a.elapsed_time(b), cnt + 1, on_main)
        import ctypes
        for arr, names, on_main in self.raw:
            for i, name in enumerate(names):
                if name is None:
                    continue
                ms = ctypes.c_float()
                if self._hip.hipEventElapsedTime(ctypes.byref(ms), ctypes.c_void_p(arr[i]), ctypes.c_void_p(arr[i + 1])) != 0:
                    self._hip.hipGetLastError()               # slot not recorded (stage not run): clear, skip
                    continue
                tot, cnt, _ = acc.get(name, (0.0, 0, on_main))
                acc[name] = (tot + ms.value, cnt + 1, on_main)
        return {k: {'total_ms': v[0], 'sampled': v[1], 'launches': self.calls.get(k, v[1]), 'avg_us': 1e3 * v[0] / v[1],
                    'main_stream': v[2]}
                for k, v in acc.items()}


def _dims(name: str):
    return [int(v) for v in name[name.index('[') + 1:-1].split('x')]


def kernel_base(name: str) -> str:
    return name.split('[')[0]


def algorithmic_work(name: str, cfg: dict):
    """(bound, units per launch, extra) for a timed span. The span name carries the launch's own sizes
    (`fps_clouds[64x16384]` = 64 clouds of 16384 points in ONE launch -- a grouped sampling launch covers
    group x 2B clouds), so the figure is per launch by construction. flops for the MFMA kernels, bytes for the
    memory-shaped ones; the sampler is a latency chain and is priced in distance evaluations.
    Per-pair figures are stated in DESIGN.md section 4."""
    sa = cfg['params']['cloud_features']['params']
    npoint = sa['npoint'][0]
    c = cfg['input_dim']
    base = kernel_base(name)
    if base == 'linear_pair':
        _, m, n, kk = _dims(name)
        return 'mfma', 4.0 * m * n * kk, {}
    if base in ('linear', 'linear+colmax'):
        m, n, kk = _dims(name)
        return 'mfma', 2.0 * m * n * kk, {}
    if base == 'head_conv_fused':
        pairs, pts = _dims(name)
        dims = [3 + cfg['params']['merge']['params']['mlp'][-1]] + list(cfg['params']['output']['params']['mlp'])   # 259: the reference's input width, not the padded 264
        return 'mfma', 2.0 * pairs * pts * sum(a * b for a, b in zip(dims[:-1], dims[1:])), {}
    if base == 'flow_embedding':
        pairs, pts, k = _dims(name)
        rows = pairs * pts * k
        return 'mfma', 2.0 * rows * (128 * 128 + 128 * 256) + 2.0 * rows * 128 * 5, {}
    if base == 'fps_clouds':        # serial chain: (npoint - 1) dependent samples per cloud, each a pass over the cloud
        clouds, n = _dims(name)
        evals = float(clouds) * (npoint - 1) * n
        return 'valu-latency', evals * FPS_FLOP_PER_EVAL, {'clouds_per_launch': clouds, 'dist_evals': evals,
                                                           'samples_per_cloud': npoint - 1}
    if base == 'sa_msg_fused':      # reads every cloud once + index list, writes 68-float rows
        clouds, n = _dims(name)
        return 'hbm', float(clouds) * (n * c * 4 + npoint * 4 + npoint * 68 * 4), {'clouds_per_launch': clouds}
    if base == 'knn_rows':
        pairs, pts, k = _dims(name)
        return 'hbm', float(pairs) * pts * (2 * 68 * 4 + k * 4), {}
    if base == 'fc':                # weights dominate: (n, k) f32 read once, m rows in and out
        (m,) = _dims(name)
        lin = cfg['params']['output']['params']['linear']
        dims = list(lin) + [8]
        per_layer = [(a * b + m * a + m * b) * 4.0 for a, b in zip(dims[:-1], dims[1:])]
        return 'hbm', sum(per_layer) / len(per_layer), {'note': 'mean over the %d fully connected launches' % len(per_layer)}
    return 'hbm', 0.0, {}


def load_traffic():
    """HBM bytes per launch from the PMC passes kept under profiles/ (a counter run cannot share a process with
    the timed region). Returned only if they were collected on the kernels this run uses."""
    try:
        with open(TRAFFIC_FILE) as fh:
            doc = json.load(fh)
    except (OSError, ValueError):
        return {}, 'no profiles/pmc_traffic.json'
    if doc.get('kernel_source_hash') != kernel_source_hash():
        return {}, 'profiles/pmc_traffic.json was collected on other kernel sources (commit {}): not attached'.format(
            doc.get('commit', '?'))
    return doc.get('configs', {}), 'profiles/pmc_traffic.json @ commit {}'.format(doc.get('commit', '?'))


def traffic_for(traffic: dict, name: str):
    """The PMC record of span `name` (`head_conv_fused[80x1024]`): the collection stores the launch size it ran at
    (`span`), and a figure taken at another launch size (--strict against the grouped run) is not this launch's figure."""
    tr = traffic.get(kernel_base(name))
    return tr if tr is not None and tr.get('span') == name else None


def cpu_baseline(cfg, sd, kind: str, points: int, budget_s: float = 12.0, pairs_cfg: int = 1):
    """The oracle (a port: the reference has no CPU path) on this host's cores: batch 1 (the reference's
    own call pattern; `value`) and, under `batched`, the configuration's batch size ('batch = 1 pair and the
    config's B (chunked if memory-bound)': at most 8 pairs per call -- the oracle materialises the grouped tensors the
    reference design does, ~0.9 GB per KITTI pair). The oracle's sampling loop parallelises over clouds, so batch 1 keeps 2 of
    the threads busy in that leg and the batched call all of them."""
    import oracle
    orc = oracle.build_oracle_model(cfg, sd)
    # a 1-GPU box grants a 16-core share of the host (more threads only oversubscribe it)
    threads = min(len(os.sched_getaffinity(0)), 16)
    torch.set_num_threads(threads)
    oracle.primitives.set_threads(threads)

    def leg(batch, budget):
        orc(torch.from_numpy(synthetic.make_batch(kind, batch, points)))             # warm-up (library init, allocator)
        calls, t0 = 0, time.perf_counter()
        while True:
            orc(torch.from_numpy(synthetic.make_batch(kind, batch, points, first_pair=(calls + 1) * batch)))
            calls += 1
            elapsed = time.perf_counter() - t0
            if elapsed > budget or calls * batch >= 64:
                break
        return calls * batch, elapsed

    done, elapsed = leg(1, budget_s)
    out = {'value': done / elapsed, 'unit': 'scan-pairs/s', 'cores': threads, 'kind': 'port',
           'sample': '{} pairs of 2x{} points, batch 1, fp32, {:.1f} s wall; torch intra-op + OpenMP threads = {}'
                     .format(done, points, elapsed, threads)}
    batch = min(pairs_cfg, 8)
    if batch > 1:
        done_b, elapsed_b = leg(batch, budget_s)
        out['batched'] = {'value': done_b / elapsed_b, 'unit': 'scan-pairs/s', 'batch': batch, 'cores': threads,
                          'sample': '{} pairs of 2x{} points in calls of {} pairs{}, fp32, {:.1f} s wall'.format(
                              done_b, points, batch, '' if batch == pairs_cfg else ' (the configuration\'s {} pairs chunked)'
                              .format(pairs_cfg), elapsed_b)}
    return out


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None)
    ap.add_argument('--warmup', type=int, default=None)
    ap.add_argument('--config', default='c2', choices=sorted(CONFIGS),
                    help='workload: c2 = BASELINE.json configs[1] (headline), c4 = configs[3], c5 = configs[4]')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--alone-only', type=int, default=0, metavar='N',
                    help='no timed window: N passes of the launches one after another (for rocprofv3)')
    ap.add_argument('--no-launch-timer', action='store_true', help='skip per-kernel HIP events (roofline = null)')
    ap.add_argument('--no-overlap', action='store_true', help='run sampling in line instead of batches ahead')
    ap.add_argument('--depth', type=int, default=None, help='batches whose sampling runs ahead on side streams')
    ap.add_argument('--sequence', action='store_true',
                    help='odometry mode (not the BASELINE metric): each step is a chunk of 16 consecutive frames of '
                         'one sequence = 16 pairs, every frame sampled and abstracted once')
    ap.add_argument('--group', type=int, default=None, help='batches sampled by one launch on a side stream')
    ap.add_argument('--dense-group', type=int, default=None, choices=[0, 1],
                    help='1: the dense stages (flow embedding, head, FC tail) of the batches sampled together also run '
                         'as one launch sequence over group x B pairs')
    ap.add_argument('--eager-dense', type=int, default=None, choices=[0, 1],
                    help='dense groups: 1 = enqueue a group\'s dense stages right behind its sampling launch (round 6 experiment: '
                         'slower, 40.8k against 42.6k in the 20-step window), 0 / default = at the step that hands out the '
                         'group\'s first batch')
    ap.add_argument('--strict', action='store_true',
                    help='no cross-batch fusion: --group 1 --dense-group 0 (every launch covers ONE batch of B pairs; '
                         'sampling still runs `depth` batches ahead on side streams)')
    ap.add_argument('--h2d', action='store_true',
                    help="each step's batch is copied from pinned host memory inside the timed loop (copy stream)")
    ap.add_argument('--clouds', default='gauss', choices=['gauss', 'ring'],
                    help='gauss: SURVEY.md 8(d) Gaussian clouds (headline); ring: LiDAR-density ring scans (KITTI configs)')
    ap.add_argument('--latency', action='store_true',
                    help='one pair per ModelInferenceHelper.predict call (pairwise and sequential), per-pair ms')
    ap.add_argument('--dense-streams', type=int, default=None,
                    help='single-batch dense launches (--strict, c4): the dense stages of consecutive batches alternate over '
                         'this many high-priority streams (1: all on the caller\'s stream; default 1, --strict 4)')
    ap.add_argument('--same-batch', action='store_true',
                    help='feed the SAME resident batch every step (rounds 1-3; scripts/timing.py:27-34 does that too) instead '
                         'of a resident ring of distinct batches: the batches of a grouped launch then alias in memory')
    ap.add_argument('--pose-budget', type=float, default=12.0,
                    help='seconds of CPU oracle time for the pose check of the last two launch groups (outside the timed region)')
    ap.add_argument('--pose-pairs', type=int, default=0,
                    help='stop the pose check after this many pairs even if not every batch of the last two groups was covered '
                         '(0: every batch gets at least one pair; the secondary passes use a few pairs each)')
    ap.add_argument('--no-cpu-leg', action='store_true',
                    help='pose check against the oracle, but no cpu_baseline timing (the secondary passes)')
    ap.add_argument('--no-secondary', action='store_true',
                    help='the plain `bench.py [--gpus 1 --steps K --warmup W]` run prints the headline line only, without the '
                         '`secondary` block (steady state, --strict, ring scans, c4, c5, single-pair latency measured by '
                         'further child processes after the headline window)')
    ap.add_argument('--cpu-stub', action='store_true',
                    help='no GPU: gloo process group and a stand-in compute function (tests of the multi-rank plumbing)')
    ap.add_argument('--gather-every', type=int, default=None,
                    help='steps whose outputs share one all-gather (N > 1); default: the dense group size, else 4')
    ap.add_argument('--force-dist', action='store_true',
                    help='initialise the RCCL process group even for one rank (exercises the all-gather path on one GPU)')
    ap.add_argument('--ahead', default='knn', choices=['sample', 'features', 'knn'], help='stages run ahead')
    args = ap.parse_args(argv)
    wl = CONFIGS[args.config]
    if args.strict:
        if args.group not in (None, 1) or args.dense_group not in (None, 0):
            ap.error('--strict fixes --group 1 --dense-group 0')
        args.group, args.dense_group = 1, 0
        # one batch per launch: 6 sampling streams and 4 dense streams measured best (on ONE stream the six
        # dependent dense launches of a batch, 0.25-0.37 ms under contention, set the pace: 30.0k pairs/s at 3 x 1, 34.5k at 6 x 4)
        if args.depth is None:
            args.depth = 6
        if args.dense_streams is None:
            args.dense_streams = 4
    if args.dense_streams is None:
        args.dense_streams = 1
    for key in ('steps', 'warmup', 'depth', 'group', 'dense_group'):
        if getattr(args, key) is None:
            setattr(args, key, wl[key])
    if args.clouds == 'ring' and wl['kind'] != 'kitti':
        ap.error('--clouds ring is a LiDAR scan: KITTI configurations (c2, c5) only')
    if args.latency and (args.gpus != 1 or args.sequence or args.h2d):
        ap.error('--latency is a single-GPU, single-pair mode')
    if args.cpu_stub:
        args.no_overlap = True                      # the stand-in has no stages to overlap
    dense = bool(args.dense_group) and args.ahead == 'knn' and args.group > 1 and not args.no_overlap and not args.sequence
    if dense and not args.latency and not args.alone_only and args.steps % args.group != 0:
        # a dense (and a sampling) launch covers `group` batches and is enqueued at a group boundary only: a window that
        # is not a whole number of groups would credit work it did not do (or do work it does not credit)
        ap.error('--steps {} is not a multiple of --group {}: with dense groups the timed window must hold whole '
                 'groups (use --steps {})'.format(args.steps, args.group, -(-args.steps // args.group) * args.group))
    return args


class OutputGather:
    """Pose outputs of `gather_every` consecutive steps share one all-gather: the steps write side by side into
    one send buffer (bigger, fewer collectives; every call makes the main stream wait for RCCL's stream)."""

    def __init__(self, dist, world: int, gather_every: int, pairs: int, dim: int, device):
        self.dist, self.every, self.pairs = dist, max(1, gather_every), pairs
        self.send = torch.zeros(self.every, pairs, dim, device=device)
        self.gathered = torch.empty(world * self.every * pairs, dim, device=device)
        self.filled = 0
        self.collectives = 0

    def slot(self):
        return self.send[self.filled]

    def put(self, y):
        """The step's outputs are in slot() already (written in place) or handed in as `y`."""
        if y is not None:
            self.send[self.filled, :y.shape[0]].copy_(y[-self.pairs:] if y.shape[0] > self.pairs else y)
        self.filled += 1
        if self.filled == self.every:
            self.flush()

    def flush(self):
        if self.filled:
            self.dist.all_gather_into_tensor(self.gathered, self.send.view(-1, self.send.shape[-1]))
            self.collectives += 1
            self.filled = 0


class StubModel:
    """--cpu-stub: stands in for the model where there is no GPU. `rows` of outputs per step whose values encode
    (rank, step), so that the all-gather's contents can be checked on every rank."""
    label_dim = 8
    npoint = 0

    def __init__(self, rank: int, pairs: int):
        self.rank, self.pairs, self.step_no = rank, pairs, 0

    def __call__(self, x):
        y = torch.full((self.pairs, self.label_dim), float(1000 * self.rank + self.step_no))
        self.step_no += 1
        return y, None, None

    @staticmethod
    def expected(rank: int, step: int) -> float:
        return float(1000 * rank + step)


def pose_deltas(y, x, cfg, sd, pairs_cfg: int, sequence: bool, rows=None):
    """max |M_hip - M_oracle| on the 4x4 for the pairs of one step (all of them, or the output rows listed in `rows`); the
    oracle runs on the host, outside the timed region."""
    import oracle
    from oracle import labels as olabels
    lt = LabelType.POSE3D_DUAL_QUAT
    orc = oracle.build_oracle_model(cfg, sd)
    x_cpu, y_cpu = x.cpu(), y.cpu().numpy()
    deltas = []
    n_out = y_cpu.shape[0]
    for row in (range(n_out) if rows is None else rows):
        if sequence:                                   # output row r pairs frames r, r + 1 of the chunk (row 0: carried frame)
            if row == 0:
                continue
            clouds = [row - 1, row]
        else:
            clouds = [row, pairs_cfg + row]
        y_ref = orc(x_cpu[clouds])
        deltas.append(float(np.abs(lt.to_matrix(y_cpu[row]) - olabels.dual_quat_to_matrix(y_ref[0].numpy())).max()))
    return deltas


def pose_check(recent, cfg, sd, pairs_cfg: int, sequence: bool, budget_s: float, max_pairs: int = 0):
    """Pose check over the steps in `recent` = [(batch, outputs)] (the last two launch groups): every pair while the CPU
    budget lasts, dealt so that every batch is covered before any batch gets a second pair (`max_pairs` > 0: at most
    that many pairs, taken from batches spread evenly over `recent`). Returns (deltas, pairs available, batches covered)."""
    total = sum(y.shape[0] for _, y in recent)
    if sequence or len(recent) == 1:
        rows = None if not max_pairs or sequence else list(range(min(max_pairs, recent[-1][1].shape[0])))
        d = pose_deltas(recent[-1][1], recent[-1][0], cfg, sd, pairs_cfg, sequence, rows=rows)
        return d, total, 1
    if max_pairs and max_pairs < len(recent):
        keep = sorted({int(round(i * (len(recent) - 1) / max(1, max_pairs - 1))) for i in range(max_pairs)})
        recent = [recent[i] for i in keep]
    deltas, covered, t0 = [], set(), time.perf_counter()
    for row in range(pairs_cfg):
        for i, (xb, yb) in enumerate(recent):
            if max_pairs and len(deltas) >= max_pairs:
                return deltas, total, len(covered)
            if time.perf_counter() - t0 > budget_s and len(covered) == len(recent):
                return deltas, total, len(covered)
            deltas += pose_deltas(yb, xb, cfg, sd, pairs_cfg, False, rows=[row])
            covered.add(i)
    return deltas, total, len(covered)


def run_latency(args, model, cfg, sd, kind, points, dev):
    """The reference's call pattern: ONE pair per call through ModelInferenceHelper.predict, timed as
    /root/reference/scripts/timing.py:32-47 does (events around predict, synchronize, per-pair ms; the clouds are on the
    device before the first event, as there)."""
    from deepclr_amd.models import ModelInferenceHelper
    x = torch.from_numpy(synthetic.make_batch(kind, 1, points)).to(dev)
    out = {}
    iters = max(args.steps, 20)
    for seq in (False, True):
        helper = ModelInferenceHelper(model, is_sequential=seq)
        if seq:
            helper.predict(x[0])
        times = []
        for i in range(args.warmup + iters):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            y = helper.predict(x[1]) if seq else helper.predict(x[1], x[0])
            b.record()
            torch.cuda.synchronize()
            if i >= args.warmup:
                times.append(a.elapsed_time(b))
        times.sort()
        out['sequential' if seq else 'pairwise'] = {
            'median_ms': times[len(times) // 2], 'min_ms': times[0], 'p90_ms': times[int(0.9 * (len(times) - 1))],
            'mean_ms': sum(times) / len(times), 'calls': len(times)}
        if not seq:
            y_pair = y
    # wall-clock throughput of back-to-back calls without a synchronize in between (what a caller that only needs the
    # poses at the end of a sequence gets)
    helper = ModelInferenceHelper(model, is_sequential=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        helper.predict(x[1], x[0])
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timer = LaunchTimer(sample_every=1)
    ops.TIMER = timer
    for _ in range(5):
        helper.predict(x[1], x[0])
    torch.cuda.synchronize()
    ops.TIMER = None
    kernels = {k: round(v['avg_us'], 1) for k, v in sorted(timer.summary().items())}
    timer.close()
    return out, y_pair, x, iters / elapsed, kernels


def run(args):
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus {} but WORLD_SIZE={}'.format(args.gpus, world))
    pinned = pin_rank(local_rank, int(os.environ.get('LOCAL_WORLD_SIZE', world)))      # before any worker thread exists
    wl = CONFIGS[args.config]
    kind, pairs_cfg, points = wl['kind'], wl['pairs'], wl['points']
    cloud_kind = 'ring' if args.clouds == 'ring' else kind
    stub = args.cpu_stub
    if stub:
        dev = torch.device('cpu')
        sync = lambda: None                                                         # noqa: E731
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device('cuda', local_rank)
        sync = torch.cuda.synchronize
        if os.environ.get('DCLR_BENCH_DENSE_PRIO', '1') != '0':
            # The caller's stream (flow embedding, head, FC tail: the stages a batch's result waits for) at high priority,
            # the runner's side streams (sampling and set abstraction batches ahead) at the default one: the head's
            # workgroups no longer queue behind set-abstraction / kNN launches for the CUs a resident sampler leaves (in-run
            # 995 -> 910 us per 80 pairs, throughput +0.7 %; DCLR_BENCH_DENSE_PRIO=0: the default stream, for A/B).
            torch.cuda.set_stream(torch.cuda.Stream(device=dev, priority=-1))
    dist = None
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch.distributed as dist
        if 'MASTER_ADDR' not in os.environ:
            os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(free_port()), RANK='0', WORLD_SIZE='1')
        if stub:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=dev)          # nccl == RCCL on ROCm

    cfg = synthetic.model_cfg(kind)
    sd = None
    if stub:
        model = StubModel(rank, pairs_cfg)
        x = torch.zeros(2 * pairs_cfg, 4, cfg['input_dim'])
        args.no_overlap, args.no_launch_timer, args.no_cpu_baseline = True, True, True
    else:
        sd = synthetic.random_state_dict(cfg, seed=0)
        model = build_model(model_config_from_dict(cfg))
        model.load_state_dict(sd)
        model = model.to(dev).eval()

    if args.latency:
        lat, y, x, back_to_back, kernels = run_latency(args, model, cfg, sd, cloud_kind, points, dev)
        med = lat['pairwise']['median_ms']
        result = {
            'metric': 'scan-pairs/sec (2x{} pts)'.format(points), 'value': 1e3 / med, 'unit': 'scan-pairs/s',
            'n_gpus': 1, 'steps': max(args.steps, 20), 'warmup': args.warmup, 'ms_per_step': med,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32 (matrix products: f16 hi/lo split operands, f32 accumulate)' if ops.PRECISION == 'f16x2' else 'f32',
            'data': 'synthetic',
            'config': {'workload': 'ONE {}-like scan pair per ModelInferenceHelper.predict call, 2x{} pts ({}; the '
                                   "reference's own call pattern, scripts/timing.py:32-47); value = 1 / median per-pair "
                                   'latency'.format(cloud_kind, points, wl['baseline']),
                       'id': args.config, 'mode': 'latency', 'clouds': args.clouds, 'pairs_per_gpu': 1,
                       'points_per_cloud': points, 'parallelism': 'dp1'},
            'latency_ms': lat, 'back_to_back_pairs_per_s': back_to_back, 'kernels_us': kernels, 'roofline': None,
        }
        if not args.no_cpu_baseline:
            result['pose_delta_vs_oracle'] = pose_deltas(y.view(1, -1), x, cfg, sd, 1, False)[0]
            if not args.no_cpu_leg:
                result['cpu_baseline'] = cpu_baseline(cfg, sd, cloud_kind, points)
        print(json.dumps(result), flush=True)
        return

    batches = None
    if not stub:
        # The timed loop walks a RESIDENT RING of distinct batches (views of one chunk, so that grouped launches read
        # their batches in place at a real stride): one more launch group than the pipeline holds, so no batch is in flight
        # twice. --same-batch: one batch every step (rounds 1-3); the batches of a grouped launch then alias in memory and
        # the sampler's / set abstraction's cloud reads hit in L2.
        n_ring = 1 if (args.same_batch or args.sequence) else \
            max(2, (1 if args.no_overlap else args.depth + 1) * args.group)
        x_host = torch.from_numpy(np.stack([
            synthetic.make_batch(cloud_kind, pairs_cfg, points, first_pair=(rank * n_ring + i) * pairs_cfg)
            for i in range(n_ring)]))
        chunk = x_host.to(dev)
        batches = [chunk[i] for i in range(n_ring)]             # the SAME view objects every lap: the runner matches by identity
        x_host = x_host[0]
        x = batches[0]
    else:
        batches = [x]

    pairs_per_step = pairs_cfg
    feeder = None
    if args.sequence:
        if args.no_overlap:
            raise SystemExit('bench.py: --sequence runs through the pipelined runner')
        pairs_per_step = x.shape[0]
        runner = PipelinedSequence(model, depth=args.depth, ahead='features' if args.ahead == 'knn' else args.ahead,
                                   group=args.group, dense_group=bool(args.dense_group) and args.group > 1)
        runner.prefetch(x)
        runner.step(x)                       # first chunk: caches the frame the timed chunks start from
    else:
        dense_group = bool(args.dense_group) and args.ahead == 'knn' and args.group > 1
        runner = None if args.no_overlap else PipelinedForward(model, depth=args.depth, ahead=args.ahead,
                                                               group=args.group, dense_group=dense_group,
                                                               dense_streams=1 if dense_group else args.dense_streams,
                                                               eager_dense=(None if args.eager_dense is None else
                                                                            bool(args.eager_dense) and dense_group),
                                                               inputs_ready=True)   # resident and never rewritten, or
                                                                                    # ordered by the feeder's copy events
    if args.h2d:
        if runner is None or args.sequence:
            raise SystemExit('bench.py: --h2d runs through the pipelined runner')
        # the loader's side: chunks of consecutive batches in pinned memory (here: the same batch, `chunk` times)
        feeder = HostBatchFeeder(runner, x)
        host_chunk = torch.stack([batches[i % len(batches)].cpu() for i in range(feeder.chunk)]).pin_memory()
        while feeder.pending() < args.depth * args.group and feeder.room():
            feeder.feed(host_chunk)
        feeder.fill()
    elif runner is not None:
        for i in range(args.depth * args.group):
            runner.prefetch(batches[i % len(batches)], flush=False)

    if args.gather_every is None:            # one all-gather per dense group: its outputs are written in place
        args.gather_every = args.group if getattr(runner, '_dense_group', False) else 4
    gather = OutputGather(dist, world, args.gather_every, pairs_per_step, model.label_dim, dev) if use_dist else None
    ranks_seen = [0]
    if use_dist:                             # proof that the collective spans `world` ranks
        mine = torch.tensor([rank], device=dev, dtype=torch.int32)
        seen = torch.empty(world, device=dev, dtype=torch.int32)
        dist.all_gather_into_tensor(seen, mine)
        ranks_seen = [int(v) for v in seen.cpu()]

    in_place_left = [0]
    stepped = [0]                                # steps taken so far = index (mod the ring) of the next batch
    recent = []                                  # (batch, outputs) of the latest steps: two launch groups for the pose check
    keep_recent = 1 if args.sequence else 2 * max(1, args.group)

    def step():
        # the outputs go straight into the all-gather's send buffer where the runner allows it: one slot per step,
        # or the slots of a whole dense group at its first step
        out = None
        if feeder is not None:
            if feeder.room() and feeder.pending() <= args.depth * args.group:
                feeder.feed(host_chunk)                        # one copy per `chunk` steps, on the copy stream
            y = feeder.step()
            recent.append((batches[(stepped[0] % feeder.chunk) % len(batches)], y))
        elif runner is not None:
            cur = batches[stepped[0] % len(batches)]
            if gather is not None and not args.sequence:
                span = runner.group_start(cur)
                if span == 0 and in_place_left[0] == 0 and not runner._dense_group:
                    out, in_place_left[0] = gather.slot(), 1
                elif span > 0 and gather.filled + span <= gather.every:
                    out = gather.send[gather.filled:gather.filled + span].view(span * pairs_per_step, -1)
                    in_place_left[0] = span
            nxt = runner.prefetched if not args.sequence else stepped[0] + 1      # the oldest batch not yet handed in
            y = runner.step(cur, upcoming=[batches[nxt % len(batches)]], out=out)
            recent.append((cur, y))
        else:
            cur = batches[stepped[0] % len(batches)]
            with torch.no_grad():
                y, _, _ = model(cur)
            recent.append((cur, y))
        stepped[0] += 1
        if len(recent) > keep_recent:
            del recent[0]
        if gather is not None:
            if in_place_left[0] > 0:
                in_place_left[0] -= 1
                gather.put(None)
            else:
                gather.put(y)
        return y

    def fence():
        if gather is not None:
            gather.flush()
            dist.barrier()
        sync()

    def solo_pass(iters):
        """Untimed pass: the same launches one after another on one stream, so that each kernel's duration is its
        own (in the timed region they share the CUs with the sampler running ahead). Returns (per-kernel summary,
        mean barrier rounds of the sampler)."""
        solo = LaunchTimer(sample_every=1, raw_pool=2 * iters + 2)
        ops.TIMER = solo
        with torch.no_grad():
            g = args.group if (args.group > 1 and not args.sequence and runner is not None) else 1
            dense_g = g if getattr(runner, '_dense_group', False) else 1
            half = x.shape[0] // 2
            grp = [batches[i % len(batches)] for i in range(g)]
            xs = torch.cat([b[:half] for b in grp] + [b[half:] for b in grp]) if dense_g > 1 else torch.cat(grp)
            for _ in range(iters):
                if args.sequence:
                    f_rows = model.cloud_feature_rows(x)
                    rows, pairs, _ = model.sequence_rows(f_rows, x.shape[0], f_rows[-model.npoint:])
                    model.merge_rows(rows, pairs)
                else:
                    # the same launch sizes as the timed region: sampling stages over group x 2B clouds, the dense
                    # stages over the batches of one dense launch
                    f_all = model.cloud_feature_rows(xs)
                    if dense_g > 1:
                        model.merge_rows(f_all, half * g)
                    else:
                        model.merge_rows(f_all[:x.shape[0] * model.npoint], half)
        torch.cuda.synchronize()
        ops.TIMER = None
        rounds = None
        if not args.sequence:
            with torch.no_grad():
                sample = model.sample(xs)                      # (idx, group_pts, group_box): box[:, 0, 6] = barrier rounds
            if sample[2] is not None:
                rounds = float(sample[2][:, 0, 6].mean())
        summary = solo.summary()
        solo.close()
        return summary, rounds

    def latency_pass(n_steps):
        """Untimed pass through the SAME pipeline: per batch, the time from the step in which the runner accepted it
        (prefetch) to the moment its pose outputs were complete on the device; a watcher thread waits on one event per step
        (recorded behind that step's outputs) and stamps the clock. The pass is CLOSED-LOOP: a step is taken only once the
        outputs of the step `outstanding` = (side streams + 1) x batches per launch earlier are complete -- a client that
        keeps the pipeline exactly full. (The timed loop is open-loop: the host enqueues hundreds of steps ahead of the GPU,
        and a batch's wait in that queue is the loop's doing, not the pipeline's.)"""
        import threading
        import queue
        if runner is None or feeder is not None or args.sequence:
            return None
        q, done_at = queue.Queue(), {}

        def watch():
            torch.cuda.set_device(dev)
            while True:
                item = q.get()
                if item is None:
                    return
                seq, ev = item
                ev.synchronize()
                done_at[seq] = time.perf_counter()
        th = threading.Thread(target=watch, daemon=True)
        th.start()
        submitted = {}
        first = stepped[0]
        outstanding = (args.depth + 1) * args.group
        for _ in range(n_steps):
            need = stepped[0] - outstanding                   # closed loop: that step's outputs must be out
            while need >= first and need not in done_at:
                time.sleep(2e-5)
            before, t = runner.prefetched, time.perf_counter()
            step()
            for seq in range(before, runner.prefetched):
                submitted[seq] = t
            ev = torch.cuda.Event()
            ev.record()
            q.put((stepped[0] - 1, ev))
        q.put(None)
        th.join()
        sync()
        lat = sorted(1e3 * (done_at[sq] - submitted[sq]) for sq in done_at if sq in submitted and sq >= first)
        if not lat:
            return None
        return {'median': lat[len(lat) // 2], 'p90': lat[int(0.9 * (len(lat) - 1))], 'max': lat[-1], 'min': lat[0],
                'batches': len(lat), 'outstanding_batches': outstanding,
                'definition': 'per batch: host time at which the runner accepted it for sampling -> its pose outputs complete '
                              'on the device (event wait in a watcher thread); untimed closed-loop pass over the same pipeline '
                              'with `outstanding_batches` batches in flight'}

    if args.alone_only:
        # profiling aid (profiles/collect.py): no timed window, only the launches one after another at the launch
        # sizes of the pipelined run -- a rocprofv3 kernel-stats summary of this command backs `alone_us`
        with torch.no_grad():                    # the first forward is the range-checked one (dense stages twice, once on the
            model(x)                             # f32 kernels): keep it out of the pass the profiler summarises
        sync()
        summary, rounds = solo_pass(args.alone_only)
        if rank == 0:
            print(json.dumps({'alone_pass': True, 'config': args.config, 'clouds': args.clouds, 'iterations': args.alone_only,
                              'kernels_alone_us': {k: round(v['avg_us'], 1) for k, v in summary.items()},
                              'fps_rounds': rounds}))
        return

    for _ in range(args.warmup):
        y = step()
    # launches that cover < 64 pairs come several per step: bracket every third (events on all of them cost ~10 %)
    timer = None if args.no_launch_timer else LaunchTimer(sample_every=3 if args.group * pairs_cfg < 64 else 1)
    ops.TIMER = timer
    copied0 = feeder.bytes_copied if feeder is not None else 0
    fence()
    host_trace = [] if os.environ.get('DCLR_BENCH_HOSTTRACE') else None    # diagnostics: host time at which each step returned
    prof = None
    if os.environ.get('DCLR_BENCH_HOSTTRACE') == '2':                      # ... and a cProfile of the timed loop (slows it)
        import cProfile
        prof = cProfile.Profile()
        prof.enable()
    mods_before = set(sys.modules) if host_trace is not None else None
    t0 = time.perf_counter()
    for _ in range(args.steps):
        y = step()
        if timer is not None:
            timer.next_step()
        if host_trace is not None:
            host_trace.append(time.perf_counter() - t0)
    fence()
    elapsed = time.perf_counter() - t0
    if prof is not None:
        import pstats
        prof.disable()
        st = pstats.Stats(prof, stream=sys.stderr)
        rows_ = sorted(st.stats.items(), key=lambda kv: -kv[1][2])[:40]       # by own time
        for (fn, line, name), (cc, nc, tt, ct, _) in rows_:
            print('%7.0f us own %7.0f us cum %5d calls  %s:%d %s' % (1e6 * tt, 1e6 * ct, nc, os.path.basename(fn), line, name), file=sys.stderr)
    if host_trace is not None and rank == 0:
        print('modules imported inside the timed loop: {}'.format(sorted(set(sys.modules) - mods_before)), file=sys.stderr)
        print('host trace (us after t0, per step): ' + ' '.join('%.0f' % (1e6 * v) for v in host_trace) +
              ' | closing fence returned at %.0f' % (1e6 * elapsed), file=sys.stderr)
    ops.TIMER = None
    if not stub:
        model.check_range()                      # the split-f16 kernels' sticky flag: a clamped activation = wrong poses = no line
    # the last two launch groups of the TIMED loop (pose check below). Copies: with an all-gather the outputs are views of its
    # send buffer, which the untimed passes below (latency_pass steps through the same closure when one rank runs with
    # --force-dist) would overwrite -- BENCH collections of round 6 read a pose delta of 0.41 on `dist1` for that reason alone
    recent_kept = [(b, yy.clone()) for b, yy in recent]
    own_block_ok = None
    if gather is not None and not stub:
        # this rank's block of the last all-gather (flushed by the closing fence) is what it sent: RCCL moved the bytes
        own_block_ok = bool(torch.equal(gather.gathered.view(world, -1)[rank], gather.send.view(-1)))
    alone, fps_rounds, latency = None, None, None
    if rank == 0 and not stub and world == 1:
        latency = latency_pass((args.depth + 4) * args.group if args.group > 1 else 5 * args.depth + 10)   # > the batches in flight
    if timer is not None and rank == 0:
        alone, fps_rounds = solo_pass(6)
    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    gather_check = None
    if stub and gather is not None:
        # every rank's block of the last flushed all-gather must hold that rank's values for the steps it covered
        # (the fence after the warm-up flushes, so slots count from the start of the timed window) -- checked ON EVERY
        # RANK (each holds the whole gathered buffer) and agreed on through a MIN all-reduce
        covered = args.steps % gather.every or gather.every
        first = args.warmup + args.steps - covered
        got = gather.gathered.view(world, gather.every, pairs_per_step, -1)
        mine = all(bool((got[r, s] == StubModel.expected(r, first + s)).all())
                   for r in range(world) for s in range(covered))
        flag = torch.tensor([1 if mine else 0], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        gather_check = bool(flag.item())

    if rank == 0:
        pairs_total = world * pairs_per_step * args.steps
        roofline, kernels, rooflines, roofline_sampler = None, None, None, None
        traffic_all, traffic_src = load_traffic()
        traffic = traffic_all.get(args.config if args.clouds == 'gauss' else args.config + '_' + args.clouds, {})
        if timer is not None:
            kernels = timer.summary()
            timer.close()

            def roof(name):
                bound, units, extra = algorithmic_work(name, cfg)
                sec = kernels[name]['avg_us'] * 1e-6
                basis = None
                if bound == 'mfma':
                    # algorithmic (f32-equivalent) FLOP of the layer shapes; the fused flow / head kernels spend
                    # SPLIT_PRODUCTS f16 MFMAs per product, so their ceiling is the f16 dense peak / SPLIT_PRODUCTS
                    split = ops.PRECISION == 'f16x2' and kernel_base(name) in ('flow_embedding', 'head_conv_fused')
                    peak = F16_MATRIX_PEAK_TFLOPS / SPLIT_PRODUCTS if split else FP32_MATRIX_PEAK_TFLOPS
                    basis = ('f16 dense MFMA peak / 3 instructions per f32-accurate product' if split
                             else 'f32 MFMA peak')
                    achieved, unit = units / sec / 1e12, 'TFLOP/s'
                elif bound == 'valu-latency':
                    peak, unit, achieved = FP32_VECTOR_PEAK_TFLOPS, 'TFLOP/s', units / sec / 1e12
                    basis = ('serial chain, no roofline applies: unpruned distance evaluations x {} FLOP against the '
                             'f32 vector peak, for scale only').format(FPS_FLOP_PER_EVAL)
                    extra = dict(extra, sample_rounds_per_s_per_cloud=extra['samples_per_cloud'] / sec,
                                 sample_rounds_per_s=extra['samples_per_cloud'] * extra['clouds_per_launch'] / sec,
                                 dist_evals_per_s=extra['dist_evals'] / sec,
                                 us_per_sample=1e6 * sec / extra['samples_per_cloud'],
                                 barrier_rounds_per_cloud=fps_rounds,
                                 samples_per_round=None if not fps_rounds else extra['samples_per_cloud'] / fps_rounds)
                else:
                    achieved, peak, unit = units / sec / 1e9, HBM_PEAK_GBS, 'GB/s'
                solo_us = alone[name]['avg_us'] if alone and name in alone else None
                tr = traffic_for(traffic, name)
                out = {'kernel': name, 'bound': bound, 'achieved': achieved, 'peak': peak, 'peak_basis': basis,
                       'unit': unit, 'frac': achieved / peak,
                       'traffic': None if tr is None else tr.get('bytes_per_launch'),
                       'avg_us': kernels[name]['avg_us'], 'alone_us': solo_us,
                       'frac_alone': None if solo_us is None else achieved / peak * kernels[name]['avg_us'] / solo_us,
                       'launches': kernels[name]['launches'], 'sampled_launches': kernels[name]['sampled'],
                       # this kernel's launches in the timed region x its average duration / the region
                       'share_of_step': kernels[name]['avg_us'] * 1e-6 * kernels[name]['launches'] / elapsed,
                       'stream': 'main' if kernels[name]['main_stream'] else 'side (overlapped)'}
                if bound == 'mfma':
                    # matrix-pipe busy fraction of the launch running alone: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x
                    # kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs (a separate --pmc pass, profiles/collect.py)
                    out['mfma_busy'] = None if tr is None else tr.get('mfma_busy')
                    out['mfma_busy_clock_ghz'] = None if tr is None else tr.get('mfma_pass_clock_ghz')
                if solo_us is not None:        # CU-time: what the launch costs the chip when it runs alone
                    out['cu_us_per_pair_alone'] = solo_us * min(1.0, _workgroups(name, cfg) / 256.0) / _pairs_in(name)
                out.update(extra)
                return out

            # `roofline` = the dominant MFMA/HBM kernel on the stream that bounds the step (the main one);
            # `roofline_sampler` = the sampler, the largest consumer of kernel time overall: a latency chain on the side
            # streams, priced as such. `rooflines` lists the eight largest of all streams.
            main = [k for k in kernels if kernels[k]['main_stream']] or list(kernels)
            weight = lambda k: kernels[k]['avg_us'] * kernels[k]['launches']            # noqa: E731
            roofline = roof(max(main, key=weight))
            fps = [k for k in kernels if kernel_base(k) == 'fps_clouds']
            if fps:
                roofline_sampler = roof(max(fps, key=weight))
            rooflines = [roof(k) for k in sorted(kernels, key=lambda k: -weight(k))[:8]]
        mode = ('sequence' if args.sequence else 'strict' if (args.group == 1 and not args.no_overlap) else
                'serial' if args.no_overlap else 'grouped')
        result = {
            'metric': 'scan-pairs/sec (2x{} pts)'.format(points), 'value': pairs_total / elapsed, 'unit': 'scan-pairs/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32 (matrix products: f16 hi/lo split operands, f32 accumulate)' if ops.PRECISION == 'f16x2' else 'f32',
            'data': 'cpu-stub: NO model ran, multi-rank plumbing only -- not a measurement' if stub else 'synthetic',
            'config': {'workload': ('odometry chunks of {0} consecutive {3}-like frames ({4} pts x {5} ch) = {1} pairs'
                                    '/GPU/step, each frame sampled once; NOT the BASELINE metric'
                                    if args.sequence else
                                    '{3}-like scan pairs, 2x{4} pts x {5} ch, {1} pairs/GPU/step ({2})')
                                   .format(x.shape[0], pairs_per_step, wl['baseline'], cloud_kind, points, x.shape[2])
                                   + '; {} architecture, seeded random weights'.format(
                                       'kitti_00-06' if kind == 'kitti' else 'modelnet40'),
                       'id': args.config, 'mode': mode, 'clouds': args.clouds,
                       'input': ('pinned host memory, copied inside the loop in chunks of {} batches'.format(feeder.chunk)
                                 if args.h2d else 'resident in HBM: ' + (
                                     'the same batch every step' if len(batches) == 1 else
                                     'a ring of {} distinct batches ({:.0f} MB), walked in order'.format(
                                         len(batches), len(batches) * x.numel() * 4 / 1e6))),
                       'pairs_per_gpu': pairs_per_step, 'points_per_cloud': points,
                       'parallelism': 'dp%d' % world,
                       'sampling_batches_ahead': 0 if runner is None else args.depth * args.group,
                       'pipeline': None if runner is None else {'side_streams': args.depth, 'batches_per_sampling_launch':
                                                                args.group, 'ahead': args.ahead,
                                                                'dense_streams': len(getattr(runner, '_dense_streams', [])) or 1,
                                                                'batches_per_dense_launch':
                                                                args.group if getattr(runner, '_dense_group', False) else 1,
                                                                'dense_enqueued': 'behind the sampling launch' if getattr(
                                                                    runner, '_eager', False) else 'at the group\'s first step'}},
            'ranks_seen': ranks_seen,
            'host_cores_of_rank0': None if not pinned else '{}-{} ({} of the {} this node grants, one disjoint slice per rank)'.format(
                pinned[0], pinned[-1], len(pinned), len(pinned) * int(os.environ.get('LOCAL_WORLD_SIZE', world))),
            'collectives': None if gather is None else {'all_gathers': gather.collectives, 'steps_per_all_gather': gather.every,
                                                         'bytes_per_rank': int(gather.send.numel() * 4)},
            'roofline': roofline,
        }
        if latency is not None:
            result['latency_ms_per_batch'] = latency
        if feeder is not None:
            copied = feeder.bytes_copied - copied0
            result['h2d'] = {'bytes_per_step': copied / args.steps, 'gb_per_s': copied / elapsed / 1e9,
                             'batches_per_copy': feeder.chunk}
        if gather_check is not None:
            result['gather_check'] = gather_check
        elif own_block_ok is not None:
            result['gather_check'] = {'own_block_of_the_last_all_gather_equals_what_was_sent': own_block_ok, 'rank': 0}
        if roofline_sampler is not None:
            result['roofline_sampler'] = roofline_sampler
        if rooflines is not None:
            result['rooflines'] = rooflines
            result['traffic_source'] = traffic_src
        if kernels is not None:
            result['kernels_us'] = {k: round(v['avg_us'], 1) for k, v in sorted(kernels.items())}
        if not args.no_cpu_baseline:
            # pose check of the last two launch groups of the timed loop against the oracle (outside the timed region), on
            # rank 0 at every world size (the metric's second half): every batch of them, as many pairs per batch as the
            # CPU budget allows. With an all-gather the outputs were written in place into its send buffer, which holds
            # the steps of the LAST gather only: older steps are not checked there.
            if gather is not None:
                recent_kept = recent_kept[-min(len(recent_kept), gather.every):]
            deltas, available, covered = pose_check(recent_kept or [(x, y)], cfg, sd, pairs_cfg, args.sequence, args.pose_budget,
                                                    max_pairs=args.pose_pairs)
            result['pose_delta_vs_oracle'] = float(np.mean(deltas))
            result['pose_delta_max'] = float(np.max(deltas))
            result['pose_delta_pairs'] = len(deltas)
            result['pose_delta_of'] = {'pairs_in_the_checked_groups': available, 'batches_covered': covered,
                                       'batches': len(recent_kept) or 1, 'rank': 0}
            if world == 1 and not args.no_cpu_leg:
                result['cpu_baseline'] = cpu_baseline(cfg, sd, cloud_kind, points, budget_s=6.0, pairs_cfg=pairs_cfg)
        print(json.dumps(result), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()



# ----------------------------------------------------------------------------------------------------------
# secondary passes: `python bench.py [--gpus 1 --steps K --warmup W]` = headline window first, then the figures
# DESIGN.md quotes beside it, each measured by a further child process and attached as `secondary`
# ----------------------------------------------------------------------------------------------------------
# name -> extra arguments of the child (each child also gets --no-secondary --no-cpu-leg and a small pose check)
SECONDARY = (
    ('steady_200', ['--config', 'c2', '--steps', '200', '--warmup', '20', '--pose-pairs', '4']),
    ('strict', ['--config', 'c2', '--strict', '--steps', '200', '--warmup', '20', '--pose-pairs', '4']),
    # plain `model(x)` on the (16, 16384, 4) batch in a loop, one stream, nothing ahead: what a caller of the REFERENCE API gets
    # at B = 8 (/root/reference/deepclr/models/deepclr.py:488-508; no PipelinedForward)
    ('serial', ['--config', 'c2', '--no-overlap', '--steps', '100', '--warmup', '10', '--pose-pairs', '4']),
    ('ring', ['--config', 'c2', '--clouds', 'ring', '--steps', '200', '--warmup', '20', '--pose-pairs', '4']),
    ('h2d', ['--config', 'c2', '--h2d', '--steps', '200', '--warmup', '20', '--pose-pairs', '4']),
    # the all-gather path on this one GPU: a one-rank RCCL group, one collective per dense group
    ('dist1', ['--config', 'c2', '--force-dist', '--steps', '200', '--warmup', '20', '--pose-pairs', '4']),
    ('sequence', ['--config', 'c2', '--sequence', '--steps', '100', '--warmup', '10', '--pose-pairs', '4']),
    ('c4', ['--config', 'c4', '--pose-pairs', '16']),
    ('c5', ['--config', 'c5', '--pose-pairs', '4']),
    ('c5_ring', ['--config', 'c5', '--clouds', 'ring', '--pose-pairs', '4']),
    ('latency', ['--config', 'c2', '--latency']),
)
SECONDARY_TIMEOUT_S = 120            # per child; a child that fails or overruns is reported as such, the headline stands
HEADLINE_TIMEOUT_S = 600
SECONDARY_DEADLINE_S = 420           # for ALL secondary passes together: past it the remaining ones are skipped (recorded as such),
                                     # so that headline + secondaries stay inside the driver's limit whatever hangs
PLAIN_FLAGS = ('--gpus', '--steps', '--warmup', '--config', '--pose-budget')


def under_profiler() -> bool:
    """rocprofv3 (or another preloaded tool) initialises the GPU inside THIS process before main() runs: such a process must
    not become the parent of GPU children (the box refuses the exec), so it measures in-process and prints the headline only."""
    env = os.environ
    return 'rocprof' in env.get('LD_PRELOAD', '').lower() or any(k.startswith(('ROCPROF', 'ROCP_', 'ROCTRACER')) for k in env)


def wants_secondary(args, argv) -> bool:
    """The plain single-GPU c2 invocation (what the driver runs) and nothing else carries the secondary block."""
    flags = [a for a in argv if a.startswith('--')]
    return (not under_profiler() and args.gpus == 1 and args.config == 'c2' and not args.no_secondary and 'WORLD_SIZE' not in os.environ
            and os.environ.get('DCLR_BENCH_CHILD') != '1' and all(f.split('=')[0] in PLAIN_FLAGS for f in flags))


def condense(line: dict) -> dict:
    """What a secondary pass contributes: throughput, step time, pose error, its dominant kernel's roofline fraction."""
    roof = line.get('roofline') or {}
    out = {'value': line.get('value'), 'unit': line.get('unit'), 'ms_per_step': line.get('ms_per_step'),
           'steps': line.get('steps'), 'warmup': line.get('warmup'), 'mode': (line.get('config') or {}).get('mode'),
           'workload': (line.get('config') or {}).get('workload'),
           'pose_delta_mean': line.get('pose_delta_vs_oracle'), 'pose_delta_max': line.get('pose_delta_max', line.get('pose_delta_vs_oracle')),
           'pose_delta_pairs': line.get('pose_delta_pairs', 1 if 'pose_delta_vs_oracle' in line else 0),
           'dominant_kernel': roof.get('kernel'), 'frac': roof.get('frac'), 'frac_alone': roof.get('frac_alone'),
           'avg_us': roof.get('avg_us'), 'alone_us': roof.get('alone_us')}
    if line.get('collectives'):
        out['ranks_seen'], out['collectives'], out['gather_check'] = line.get('ranks_seen'), line['collectives'], line.get('gather_check')
    if line.get('h2d'):
        out['h2d'] = line['h2d']
    if 'latency_ms' in line:
        out['latency_ms'] = {k: v.get('median_ms') for k, v in line['latency_ms'].items()}
        out['kernels_us'] = line.get('kernels_us')
    if 'latency_ms_per_batch' in line:
        out['latency_ms_per_batch_median'] = line['latency_ms_per_batch'].get('median')
    if line.get('roofline_sampler'):
        rs = line['roofline_sampler']
        out['sampler'] = {'kernel': rs.get('kernel'), 'avg_us': rs.get('avg_us'), 'alone_us': rs.get('alone_us'),
                          'samples_per_round': rs.get('samples_per_round')}
    return out


def _run_child(cmd, env, timeout_s):
    """subprocess.run with the child in a process group of its own, killed as a GROUP on timeout (a bench child may have
    started ranks or helper processes of its own)."""
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, start_new_session=True)
    try:
        out, _ = proc.communicate(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(proc.pid, signal.SIGKILL)
        except (ProcessLookupError, PermissionError):
            pass
        proc.communicate()
        raise
    return subprocess.CompletedProcess(cmd, proc.returncode, out, None)


def child_line(argv, timeout_s, run=None):
    """One child `bench.py argv` (it initialises the GPU; this process never does). Returns (json line | None, error | None).
    run: a stand-in for subprocess.run (tests); None = the process-group runner above."""
    env = dict(os.environ, DCLR_BENCH_CHILD='1')
    t0 = time.time()
    cmd = [sys.executable, os.path.abspath(__file__)] + list(argv)
    try:
        if run is None:
            res = _run_child(cmd, env, timeout_s)
        else:
            res = run(cmd, env=env, stdout=subprocess.PIPE, text=True, timeout=timeout_s)
    except subprocess.TimeoutExpired:
        return None, 'timed out after {} s'.format(timeout_s)
    lines = [l for l in (res.stdout or '').splitlines() if l.lstrip().startswith('{')]
    if res.returncode != 0 or not lines:
        return None, 'exit code {} after {:.0f} s, json line seen: {}'.format(res.returncode, time.time() - t0, bool(lines))
    try:
        return json.loads(lines[-1]), None
    except ValueError as exc:
        return None, 'unparsable line: {}'.format(exc)


def run_with_secondary(argv, run=None, plan=SECONDARY, out=sys.stdout, deadline_s=SECONDARY_DEADLINE_S, clock=time.time) -> int:
    """Headline first (the unchanged window, a child of its own on the idle chip), then the secondary passes one after
    another; ONE JSON line = the headline's with `secondary` attached. Any secondary failure is recorded in its slot; once
    the passes together have used `deadline_s` the remaining ones are skipped, so the line is out in bounded time."""
    head, err = child_line(list(argv) + ['--no-secondary'], HEADLINE_TIMEOUT_S, run)
    if head is None:
        sys.stderr.write('bench.py: headline run failed: {}\n'.format(err))
        return 1
    sys.stderr.write('bench.py: headline {:.1f} {} measured; secondary passes follow\n'.format(head.get('value', 0.0), head.get('unit', '')))
    t0 = clock()
    secondary = {}
    for name, extra in plan:
        left = deadline_s - (clock() - t0)
        if left < 15:
            secondary[name] = {'error': 'skipped: the secondary passes had used their {} s'.format(deadline_s), 'args': ' '.join(extra)}
            continue
        line, err = child_line(['--gpus', '1', '--no-secondary', '--no-cpu-leg'] + list(extra), min(SECONDARY_TIMEOUT_S, left), run)
        secondary[name] = {'error': err, 'args': ' '.join(extra)} if line is None else dict(condense(line), args=' '.join(extra))
    secondary['note'] = ('each entry: a separate process after the headline window, same kernels and inputs policy (resident ring '
                         'of distinct batches), untimed for `value`; pose deltas against the CPU oracle on a few pairs each; '
                         '{:.0f} s for all of them'.format(clock() - t0))
    head['secondary'] = secondary
    out.write(json.dumps(head) + '\n')
    out.flush()
    return 0


def _pairs_in(name: str) -> float:
    """Scan pairs one launch of this span serves (clouds / 2 for the per-cloud stages)."""
    d = _dims(name)
    base = kernel_base(name)
    if base in ('fps_clouds', 'sa_msg_fused'):
        return d[0] / 2.0
    if base == 'linear_pair':
        return 1.0
    return float(d[0])


def _workgroups(name: str, cfg: dict) -> float:
    """Workgroup count of the launch where it is below a full chip (the sampler: one per cloud); else 256."""
    return float(_dims(name)[0]) if kernel_base(name) == 'fps_clouds' else 256.0


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus < 1:
        raise SystemExit('bench.py: --gpus must be >= 1')
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # launched as `python bench.py --gpus N`: become the launcher (no GPU call has been made in this process)
        raise SystemExit(spawn_ranks(args.gpus, argv))
    if wants_secondary(args, argv):
        # no GPU call has been made in this process and none will be: every measurement is a child
        raise SystemExit(run_with_secondary(argv))
    run(args)


if __name__ == '__main__':
    main()
