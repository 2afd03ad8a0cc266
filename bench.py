#!/usr/bin/env python3
"""Throughput of the DeepCLR forward hot path on MI355X: scan-pairs/s (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

One step = one forward pass of the whole hot path (FPS -> set abstraction -> kNN -> flow embedding ->
pose head) over one batch of synthetic KITTI-sized pairs already resident in HBM: BASELINE.json
configs[1], 8 pairs of 2 x 16384 points per GPU. Ranks own independent pairs (weak scaling, no
data-path collective); the only exchange is an RCCL all-gather of the (8, 8) pose outputs per step.
Rank 0 prints ONE JSON line. The CPU oracle is used only for the `cpu_baseline` leg and the pose
check -- never inside the timed region.
"""
import argparse
import json
import os
import sys
import time

# The HIP runtime multiplexes all streams of a process onto 4 hardware queues by default. The runner uses the
# main stream + 3 side streams, and RCCL brings its own stream for the all-gather: with 4 queues that stream
# shares a queue with a ~1 ms sampling kernel and every step waits for it (measured on one GPU with a
# one-rank process group: 18.6k instead of 28.9k pairs/s). Must be set before the runtime initialises.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

import numpy as np                                          # noqa: E402
import torch                                                # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from deepclr_amd import ops, synthetic                      # noqa: E402
from deepclr_amd.config import model_config_from_dict       # noqa: E402
from deepclr_amd.labels import LabelType                    # noqa: E402
from deepclr_amd.models import build_model                  # noqa: E402
from deepclr_amd.pipeline import PipelinedForward, PipelinedSequence           # noqa: E402

PAIRS_PER_GPU = 8
POINTS = 16384
FP32_MATRIX_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, = fp32 vector peak
F16_MATRIX_PEAK_TFLOPS = 2516.6      # MI355X_MICROARCH.md: dense f16/bf16 MFMA = 16 x the f32 matrix rate (~2.5 PF)
SPLIT_PRODUCTS = 3                   # f16 MFMAs per f32-accurate product on the split path (csrc/mma16f.h)
HBM_PEAK_GBS = 8000.0                # MI355X_MICROARCH.md: HBM3E spec peak


class LaunchTimer:
    """HIP events around every library launch, on the stream the kernel is enqueued on."""

    SAMPLE_EVERY = 5       # bracket the launches of every 5th step only (the events themselves cost ~10 %); coprime
                           # with the 2-4 batches per sampling launch, or those launches would never be sampled

    def __init__(self, sample_every=None):
        self.spans = []
        self.raw = []
        self._hip = None
        self.main_stream = torch.cuda.current_stream().cuda_stream
        self.step = 0
        if sample_every is not None:
            self.SAMPLE_EVERY = sample_every

    def next_step(self):
        self.step += 1

    def begin(self, name):
        if self.step % self.SAMPLE_EVERY != 0:
            return None
        start = torch.cuda.Event(enable_timing=True)
        start.record()
        return (name, start, torch.cuda.current_stream().cuda_stream == self.main_stream)

    def end(self, token):
        if token is None:
            return
        stop = torch.cuda.Event(enable_timing=True)
        stop.record()
        self.spans.append((token[0], token[1], stop, token[2]))

    def merge_events(self, rows, n_fc=3):
        """Raw HIP events for the stages of one dclr_merge_forward call (the dense stages are one foreign call;
        the library records these between its launches, on the launch stream)."""
        if self.step % self.SAMPLE_EVERY != 0:
            return None
        import ctypes
        from deepclr_amd import lib
        if self._hip is None:
            self._hip = ctypes.CDLL('libamdhip64.so')
        arr = (ctypes.c_void_p * lib.MERGE_EVENTS)()
        for i in range(lib.MERGE_EVENTS):
            ev = ctypes.c_void_p()
            if self._hip.hipEventCreate(ctypes.byref(ev)) != 0:
                return None
            arr[i] = ev.value
        names = ['linear_pair[2x%dx128x64]' % rows, None, 'knn_rows', 'flow_embedding', 'head_conv_fused'] + ['fc'] * n_fc
        on_main = torch.cuda.current_stream().cuda_stream == self.main_stream
        self.raw.append((arr, names, on_main))
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
        return {k: {'total_ms': v[0], 'launches': v[1], 'avg_us': 1e3 * v[0] / v[1], 'main_stream': v[2]}
                for k, v in acc.items()}


def algorithmic_work(name: str, cfg: dict, pairs: int, n_points: int, clouds: int):
    """(bound, units) per launch: flops for the MFMA kernels, bytes for the memory-shaped ones.
    Figures are stated per scan pair in DESIGN.md section 'Kernels and rooflines'."""
    sa = cfg['params']['cloud_features']['params']
    npoint, k = sa['npoint'][0], cfg['params']['merge']['params']['k']
    c = cfg['input_dim']
    if name.startswith('linear_pair'):
        m, n, kk = (int(v) for v in name[name.index('[') + 3:-1].split('x'))
        return 'mfma', 4.0 * m * n * kk
    if name.startswith('linear'):
        m, n, kk = (int(v) for v in name[name.index('[') + 1:-1].split('x'))
        return 'mfma', 2.0 * m * n * kk
    if name == 'head_conv_fused':
        out = cfg['params']['output']['params']['mlp']
        dims = [264] + list(out)
        return 'mfma', 2.0 * pairs * npoint * sum(a * b for a, b in zip(dims[:-1], dims[1:]))
    if name == 'flow_embedding':
        rows = pairs * npoint * k
        return 'mfma', 2.0 * rows * (128 * 128 + 128 * 256) + 2.0 * rows * 128 * 5
    if name == 'fps_clouds':        # reads every cloud once, writes the indices
        return 'hbm', clouds * (n_points * c * 4 + npoint * 4)
    if name == 'sa_msg_fused':      # reads every cloud once + index list, writes 68-float rows
        return 'hbm', clouds * (n_points * c * 4 + npoint * 4 + npoint * 68 * 4)
    if name == 'knn_rows':
        return 'hbm', pairs * npoint * (2 * 68 * 4 + k * 4)
    return 'hbm', 0.0


def cpu_baseline(cfg, sd, budget_s: float = 15.0):
    """The oracle (a port: the reference has no CPU path) on this host's cores."""
    import oracle
    orc = oracle.build_oracle_model(cfg, sd)
    # a 1-GPU box grants a 16-core share of the host (more threads only oversubscribe it)
    threads = min(len(os.sched_getaffinity(0)), 16)
    torch.set_num_threads(threads)
    oracle.primitives.set_threads(threads)
    x = torch.from_numpy(synthetic.make_batch('kitti', 1, POINTS))
    orc(x)                                           # warm-up (library init, allocator)
    done, t0 = 0, time.perf_counter()
    while True:
        orc(torch.from_numpy(synthetic.make_batch('kitti', 1, POINTS, first_pair=done + 1)))
        done += 1
        elapsed = time.perf_counter() - t0
        if elapsed > budget_s or done >= 64:
            break
    return {'value': done / elapsed, 'unit': 'scan-pairs/s', 'cores': threads, 'kind': 'port',
            'sample': '{} pairs of 2x{} points, batch 1, fp32, {:.1f} s wall; torch intra-op + OpenMP threads = {}'
                      .format(done, POINTS, elapsed, threads)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-launch-timer', action='store_true', help='skip per-kernel HIP events (roofline = null)')
    ap.add_argument('--no-overlap', action='store_true', help='run sampling in line instead of batches ahead')
    ap.add_argument('--depth', type=int, default=3, help='batches whose sampling runs ahead on side streams')
    ap.add_argument('--sequence', action='store_true',
                    help='odometry mode (not the BASELINE metric): each step is a chunk of 16 consecutive frames of '
                         'one sequence = 16 pairs, every frame sampled and abstracted once')
    ap.add_argument('--group', type=int, default=4, help='batches sampled by one launch on a side stream')
    ap.add_argument('--gather-every', type=int, default=4, help='steps whose outputs share one all-gather (N > 1)')
    ap.add_argument('--force-dist', action='store_true',
                    help='initialise the RCCL process group even for one rank (exercises the all-gather path on one GPU)')
    ap.add_argument('--ahead', default='knn', choices=['sample', 'features', 'knn'], help='stages run ahead')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus {} but WORLD_SIZE={} (launch N>1 through torch.distributed.run)'
                         .format(args.gpus, world))
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    dist = None
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch.distributed as dist
        if 'MASTER_ADDR' not in os.environ:
            os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29517', RANK='0', WORLD_SIZE='1')
        dist.init_process_group('nccl', device_id=dev)          # nccl == RCCL on ROCm

    cfg = synthetic.model_cfg('kitti')
    sd = synthetic.random_state_dict(cfg, seed=0)
    model = build_model(model_config_from_dict(cfg))
    model.load_state_dict(sd)
    model = model.to(dev).eval()
    x = torch.from_numpy(synthetic.make_batch('kitti', PAIRS_PER_GPU, POINTS, first_pair=rank * PAIRS_PER_GPU)).to(dev)
    
    pairs_per_step = PAIRS_PER_GPU
    if args.sequence:
        if args.no_overlap:
            raise SystemExit('bench.py: --sequence runs through the pipelined runner')
        pairs_per_step = x.shape[0]
        runner = PipelinedSequence(model, depth=args.depth, ahead='features' if args.ahead == 'knn' else args.ahead,
                                   group=args.group)
        runner.prefetch(x)
        runner.step(x)                       # first chunk: caches the frame the timed chunks start from
    else:
        runner = None if args.no_overlap else PipelinedForward(model, depth=args.depth, ahead=args.ahead,
                                                               group=args.group)
    if runner is not None:
        for _ in range(args.depth * args.group):
            runner.prefetch(x, flush=False)

    # all-gather once per GATHER_EVERY steps: the steps' outputs are written side by side into one send buffer
    # (bigger, fewer collectives; every call makes the main stream wait for RCCL's stream)
    gather_every = max(1, args.gather_every)
    send = torch.zeros(gather_every, pairs_per_step, 8, device=dev) if use_dist else None
    gathered = torch.empty(world * gather_every * pairs_per_step, 8, device=dev) if use_dist else None
    counter = [0]

    def flush():
        dist.all_gather_into_tensor(gathered, send.view(-1, 8))
        counter[0] = 0

    def step():
        slot = send[counter[0]] if use_dist and runner is not None and not args.sequence else None
        if runner is not None:
            y = runner.step(x, upcoming=[x], out=slot)   # sampling of later batches overlaps the stages of this one
        else:
            with torch.no_grad():
                y, _, _ = model(x)
        if use_dist:
            if slot is None:
                send[counter[0], :y.shape[0]].copy_(y[-pairs_per_step:] if y.shape[0] > pairs_per_step else y)
            counter[0] += 1
            if counter[0] == gather_every:
                flush()
        return y

    def fence():
        if use_dist:
            if counter[0]:
                flush()
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        y = step()
    timer = None if args.no_launch_timer else LaunchTimer()
    ops.TIMER = timer
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        y = step()
        if timer is not None:
            timer.next_step()
    fence()
    elapsed = time.perf_counter() - t0
    ops.TIMER = None
    alone = None
    if timer is not None and rank == 0:
        # second, untimed pass: the same launches one after another on one stream, so that each kernel's
        # duration is its own (in the timed region they share the CUs with the sampler running ahead)
        solo = LaunchTimer(sample_every=1)
        ops.TIMER = solo
        with torch.no_grad():
            for _ in range(6):
                if args.sequence:
                    f_rows = model.cloud_feature_rows(x)
                    rows, pairs, _ = model.sequence_rows(f_rows, x.shape[0], f_rows[-model.npoint:])
                    model.merge_rows(rows, pairs)
                else:
                    model(x)
        torch.cuda.synchronize()
        ops.TIMER = None
        alone = solo.summary()
    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    result = None
    if rank == 0:
        pairs_total = world * pairs_per_step * args.steps
        roofline, kernels = None, None
        rooflines = None
        if timer is not None:
            kernels = timer.summary()
            sampled_steps = len(range(0, args.steps, LaunchTimer.SAMPLE_EVERY))

            traffic = {}
            try:        # HBM bytes per launch from the PMC passes kept under profiles/ (collected offline: a
                        # counter run cannot share a process with the timed region)
                with open(os.path.join(ROOT, 'profiles', 'r01_pmc_traffic.json')) as fh:
                    traffic = {k: v['bytes_per_launch_raw'] for k, v in json.load(fh)['kernels'].items()}
            except (OSError, KeyError, ValueError):
                pass

            def roof(name):
                bound, units = algorithmic_work(name, cfg, pairs_per_step, POINTS, x.shape[0])
                sec = kernels[name]['avg_us'] * 1e-6
                basis = None
                if bound == 'mfma':
                    # algorithmic (f32-equivalent) FLOP of the layer shapes; the fused flow / head kernels spend
                    # SPLIT_PRODUCTS f16 MFMAs per product, so their ceiling is the f16 dense peak / SPLIT_PRODUCTS
                    split = ops.PRECISION == 'f16x2' and name in ('flow_embedding', 'head_conv_fused')
                    peak = F16_MATRIX_PEAK_TFLOPS / SPLIT_PRODUCTS if split else FP32_MATRIX_PEAK_TFLOPS
                    basis = ('f16 dense MFMA peak / 3 instructions per f32-accurate product' if split
                             else 'f32 MFMA peak')
                    achieved, unit = units / sec / 1e12, 'TFLOP/s'
                else:
                    achieved, peak, unit = units / sec / 1e9, HBM_PEAK_GBS, 'GB/s'
                solo_us = alone[name]['avg_us'] if alone and name in alone else None
                return {'kernel': name, 'bound': bound, 'achieved': achieved, 'peak': peak, 'peak_basis': basis,
                        'unit': unit, 'frac': achieved / peak, 'traffic': traffic.get(name), 'avg_us': kernels[name]['avg_us'],
                        'alone_us': solo_us,
                        'frac_alone': None if solo_us is None else achieved / peak * kernels[name]['avg_us'] / solo_us,
                        'share_of_step': (kernels[name]['total_ms'] / max(1, sampled_steps)) / (1e3 * elapsed / args.steps),
                        'stream': 'main' if kernels[name]['main_stream'] else 'side (overlapped)'}

            # dominant kernel = largest total time on the stream that bounds the step (the main one);
            # the side-stream sampler is latency-bound by construction and listed in `rooflines`
            main = [k for k in kernels if kernels[k]['main_stream']] or list(kernels)
            roofline = roof(max(main, key=lambda k: kernels[k]['total_ms']))
            rooflines = [roof(k) for k in sorted(kernels, key=lambda k: -kernels[k]['total_ms'])[:6]]
        # pose check of the last step's first pair against the oracle (outside the timed region)
        result = {
            'metric': 'scan-pairs/sec (2x16384 pts)', 'value': pairs_total / elapsed, 'unit': 'scan-pairs/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32 (matrix products: f16 hi/lo split operands, f32 accumulate)' if ops.PRECISION == 'f16x2' else 'f32', 'data': 'synthetic',
            'config': {'workload': ('odometry chunks of {} consecutive KITTI-sized frames (16384 pts x 4 ch) = {} pairs'
                                    '/GPU/step, each frame sampled once; NOT the BASELINE metric'
                                    if args.sequence else
                                    'KITTI-sized scan pairs, 2x16384 pts x 4 ch, {1} pairs/GPU/step '
                                    '(BASELINE.json configs[1])').format(x.shape[0], pairs_per_step)
                                   + '; kitti_00-06 architecture, seeded random weights',
                       'pairs_per_gpu': pairs_per_step, 'points_per_cloud': POINTS, 'parallelism': 'dp%d' % world,
                       'sampling_batches_ahead': 0 if runner is None else args.depth * args.group,
                       'pipeline': None if runner is None else {'side_streams': args.depth, 'batches_per_sampling_launch':
                                                                args.group, 'ahead': args.ahead}},
            'roofline': roofline,
        }
        if rooflines is not None:
            result['rooflines'] = rooflines
        if kernels is not None:
            result['kernels_us'] = {k: round(v['avg_us'], 1) for k, v in sorted(kernels.items())}
        if world == 1 and not args.no_cpu_baseline:
            import oracle
            from oracle import labels as olabels
            first = [0, 1] if args.sequence else [0, PAIRS_PER_GPU]          # clouds of output row `row`
            row = 1 if args.sequence else 0
            y_ref = oracle.build_oracle_model(cfg, sd)(x[first].cpu())
            lt = LabelType.POSE3D_DUAL_QUAT
            result['pose_delta_vs_oracle'] = float(np.abs(lt.to_matrix(y[row].cpu().numpy())
                                                          - olabels.dual_quat_to_matrix(y_ref[0].numpy())).max())
            result['cpu_baseline'] = cpu_baseline(cfg, sd)
        print(json.dumps(result), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
