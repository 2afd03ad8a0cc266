"""bench.py without a GPU: the algorithmic work it prices kernels with, the launcher it becomes for --gpus N, and
the fields of its JSON contract."""
import importlib.util
import json
import os
import subprocess
import sys

import pytest
import torch

from deepclr_amd import synthetic

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def bench():
    spec = importlib.util.spec_from_file_location('bench_under_test', os.path.join(ROOT, 'bench.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_algorithmic_work_matches_the_design_figures(bench):
    cfg = synthetic.model_cfg('kitti')
    pairs = 8
    bound, flops, _ = bench.algorithmic_work('flow_embedding[8x1024x20]', cfg)
    rows = pairs * 1024 * 20
    assert bound == 'mfma' and flops == 2.0 * rows * (128 * 128 + 128 * 256) + 2.0 * rows * 128 * 5
    assert abs(flops / pairs - 2.01e9) < 0.05e9                       # 2.01 GFLOP per pair (+ layer-1 rest)
    bound, flops, _ = bench.algorithmic_work('head_conv_fused[8x1024]', cfg)
    assert bound == 'mfma' and flops / pairs == 2.0 * 1024 * 1049344      # 1024 x 1,049,344 MAC: K = 259 in layer 1, not the padded 264
    bound, flops, _ = bench.algorithmic_work('linear_pair[2x8192x128x64]', cfg)
    assert bound == 'mfma' and flops == 4.0 * 8192 * 128 * 64
    bound, nbytes, _ = bench.algorithmic_work('knn_rows[8x1024x20]', cfg)
    assert bound == 'hbm' and nbytes == 8 * 1024 * (2 * 68 * 4 + 20 * 4)
    bound, nbytes, extra = bench.algorithmic_work('fc[8]', cfg)
    layers = [(1024, 512), (512, 256), (256, 8)]
    assert bound == 'hbm' and nbytes == sum((a * b + 8 * a + 8 * b) * 4.0 for a, b in layers) / 3 and nbytes > 0


def test_grouped_launches_are_priced_per_launch(bench):
    """One sampling / set-abstraction launch of the pipelined run covers group x 2B clouds (c2: 4 x 16 = 64): the
    span name carries the launch's own size, so the figures scale with it (they were 4x low)."""
    cfg = synthetic.model_cfg('kitti')
    b16 = bench.algorithmic_work('sa_msg_fused[16x16384]', cfg)
    b64 = bench.algorithmic_work('sa_msg_fused[64x16384]', cfg)
    assert b16[0] == 'hbm' and b16[1] == 16 * (16384 * 16 + 1024 * 4 + 1024 * 68 * 4) and b64[1] == 4 * b16[1]
    assert b64[2]['clouds_per_launch'] == 64
    bound, flop, extra = bench.algorithmic_work('fps_clouds[64x16384]', cfg)
    assert bound == 'valu-latency'                                             # a latency chain, not an HBM stream
    assert extra['dist_evals'] == 64 * 1023 * 16384 and flop == extra['dist_evals'] * bench.FPS_FLOP_PER_EVAL
    assert extra['samples_per_cloud'] == 1023 and extra['clouds_per_launch'] == 64
    # ModelNet architecture: npoint 512, k 30
    mcfg = synthetic.model_cfg('modelnet')
    _, flop, extra = bench.algorithmic_work('fps_clouds[512x2048]', mcfg)
    assert extra['dist_evals'] == 512 * 511 * 2048
    _, flops, _ = bench.algorithmic_work('flow_embedding[256x512x30]', mcfg)
    # SURVEY 8(d) counts 1.013 GMAC per ModelNet pair with layer 1 applied per neighbour row (131 -> 128); the fused
    # design evaluates the two feature blocks of layer 1 once per POINT (linear_pair), leaving 5 columns per row
    assert flops / 256 == 2.0 * 512 * 30 * (128 * 128 + 128 * 256 + 128 * 5)


def test_peaks_and_workload_constants(bench):
    assert bench.PAIRS_PER_GPU == 8 and bench.POINTS == 16384                  # BASELINE.json configs[1]
    assert bench.CONFIGS['c4']['pairs'] == 256 and bench.CONFIGS['c4']['points'] == 2048      # configs[3]
    assert bench.CONFIGS['c5']['pairs'] == 4 and bench.CONFIGS['c5']['points'] == 65536       # configs[4]
    assert bench.FP32_MATRIX_PEAK_TFLOPS == 157.3 and bench.HBM_PEAK_GBS == 8000.0
    assert abs(bench.F16_MATRIX_PEAK_TFLOPS / bench.SPLIT_PRODUCTS - 838.9) < 0.1
    assert os.environ.get('GPU_MAX_HW_QUEUES') is not None                     # set on import, before HIP initialises
    args = bench.parse_args([])
    assert (args.config, args.gpus, args.group, args.depth) == ('c2', 1, 10, 3) and args.steps >= 100
    assert 20 % args.group == 0                  # the driver's 20-step window holds whole sampling / dense groups
    args = bench.parse_args(['--config', 'c4', '--steps', '7'])
    assert (args.group, args.depth, args.steps) == (1, 2, 7)


def test_measurement_modes_and_the_whole_groups_rule(bench, capsys):
    """--strict / --h2d / --clouds ring / --latency beside the headline; with dense groups a timed window that is not a
    whole number of groups would credit work it did not do: rejected."""
    args = bench.parse_args(['--strict', '--steps', '7'])
    assert (args.group, args.dense_group, args.depth, args.dense_streams, args.steps) == (1, 0, 6, 4, 7)
    assert bench.parse_args([]).dense_streams == 1 and bench.parse_args(['--strict', '--depth', '3']).depth == 3
    with pytest.raises(SystemExit):
        bench.parse_args(['--strict', '--group', '4'])
    for bad in (['--steps', '25', '--warmup', '5'], ['--config', 'c5', '--steps', '30']):
        with pytest.raises(SystemExit):
            bench.parse_args(bad)
        assert 'whole' in capsys.readouterr().err
    assert bench.parse_args(['--steps', '30']).steps == 30                        # c2: groups of 10
    assert bench.parse_args(['--steps', '25', '--dense-group', '0']).steps == 25   # no dense groups: any window
    assert bench.parse_args(['--steps', '25', '--no-overlap']).steps == 25
    assert bench.parse_args(['--steps', '20', '--warmup', '5']).warmup == 5        # the driver's arguments
    args = bench.parse_args(['--clouds', 'ring', '--h2d'])
    assert args.clouds == 'ring' and args.h2d
    with pytest.raises(SystemExit):
        bench.parse_args(['--config', 'c4', '--clouds', 'ring'])                   # a LiDAR scan is not a ModelNet object
    assert bench.parse_args(['--latency', '--steps', '25']).latency                # no group rule: one pair per call
    with pytest.raises(SystemExit):
        bench.parse_args(['--latency', '--gpus', '2'])


def test_two_real_ranks_end_to_end_on_cpu(tmp_path):
    """`python bench.py --gpus 2 --cpu-stub`: the launcher starts two real children, each runs run() on gloo with the
    stand-in compute function -- rank > 0 code, the all-gather bookkeeping, the max-over-ranks timing and rank 0's JSON
    line are exercised before any multi-GPU hardware is."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    for argv, gathers in ((['--steps', '10', '--warmup', '3', '--strict'], 4),       # 1 (warm-up fence) + 2 full + 1 partial
                          (['--steps', '8', '--warmup', '0', '--gather-every', '3'], 3)):
        out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--cpu-stub'] + argv,
                             env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
        assert len(lines) == 1                                                     # rank 0 alone prints
        line = json.loads(lines[0])
        assert line['ranks_seen'] == [0, 1] and line['n_gpus'] == 2 and line['gather_check'] is True
        assert line['collectives']['all_gathers'] == gathers and line['config']['parallelism'] == 'dp2'
        assert 'stub' in line['data'] and line['scaling'] == 'weak' and line['roofline'] is None


def test_launch_timer_samples_per_kind_not_per_step(bench, monkeypatch):
    """Launches that cover several batches (one per `group` steps) must be sampled whatever the warm-up count."""
    class _Ev:
        def __init__(self, enable_timing=True): pass
        def record(self): pass
    class _S:
        cuda_stream = 7
    monkeypatch.setattr(bench.torch.cuda, 'Event', _Ev)
    monkeypatch.setattr(bench.torch.cuda, 'current_stream', lambda: _S())
    t = bench.LaunchTimer(sample_every=3)
    hits = [t.begin('fps_clouds[64x16384]') is not None for _ in range(9)]
    assert hits == [True, False, False] * 3 and t.calls['fps_clouds[64x16384]'] == 9
    assert t.begin('sa_msg_fused[64x16384]') is not None                      # kinds are counted separately
    every = bench.LaunchTimer()                                               # default: every launch (few, grouped launches)
    assert all(every.begin('fps_clouds[160x16384]') is not None for _ in range(4))
    every.close()                                                             # no device here: no raw events to destroy
    assert every._raw_all == []


def test_rank_environments_of_the_self_launcher(bench):
    envs = bench.rank_environments(4, 23456, base_env={'PATH': '/bin', 'WORLD_SIZE': 'stale'})
    assert [e['RANK'] for e in envs] == ['0', '1', '2', '3'] and [e['LOCAL_RANK'] for e in envs] == ['0', '1', '2', '3']
    assert all(e['WORLD_SIZE'] == '4' and e['MASTER_ADDR'] == '127.0.0.1' and e['MASTER_PORT'] == '23456' for e in envs)
    assert all(e['HSA_ENABLE_IPC_MODE_LEGACY'] == '0' and e['PATH'] == '/bin' for e in envs)
    assert 0 < bench.free_port() < 65536


def test_self_launcher_spawns_before_any_gpu_call_and_relays_rank0(bench, tmp_path, monkeypatch):
    """`python bench.py --gpus 2` without WORLD_SIZE: two children with rank environments, rank 0's JSON relayed,
    a failing rank makes the launcher fail. The children here are a stub script (no GPU in this container)."""
    stub = tmp_path / 'stub_bench.py'
    stub.write_text(
        "import json, os, sys\n"
        "r = int(os.environ['RANK'])\n"
        "assert os.environ['WORLD_SIZE'] == '2' and os.environ['LOCAL_RANK'] == str(r) and os.environ['DCLR_BENCH_CHILD'] == '1'\n"
        "if '--fail' in sys.argv and r == 1: sys.exit(3)\n"
        "if r == 0: print(json.dumps({'metric': 'stub', 'n_gpus': 2, 'argv': sys.argv[1:]}))\n")
    monkeypatch.setattr(bench.os.path, 'abspath', lambda p: str(stub) if p == bench.__file__ else os.path.normpath(os.path.join(os.getcwd(), p)))
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    import io
    import contextlib
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        rc = bench.spawn_ranks(2, ['--gpus', '2', '--steps', '3'])
    assert rc == 0
    line = json.loads(buf.getvalue().strip().splitlines()[-1])
    assert line['n_gpus'] == 2 and line['argv'] == ['--gpus', '2', '--steps', '3']
    with contextlib.redirect_stdout(io.StringIO()):
        assert bench.spawn_ranks(2, ['--gpus', '2', '--fail']) == 3


def test_main_becomes_the_launcher_only_without_world_size(bench, monkeypatch):
    calls = []
    monkeypatch.setattr(bench, 'spawn_ranks', lambda n, argv: calls.append((n, list(argv))) or 0)
    monkeypatch.setattr(bench, 'run', lambda args: calls.append(('run', args.gpus)))
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main(['--gpus', '8', '--steps', '10', '--warmup', '1'])
    assert e.value.code == 0 and calls == [(8, ['--gpus', '8', '--steps', '10', '--warmup', '1'])]
    monkeypatch.setenv('WORLD_SIZE', '8')                  # under torch.distributed.run: a rank, not a launcher
    bench.main(['--gpus', '8'])
    assert calls[-1] == ('run', 8)
    monkeypatch.delenv('WORLD_SIZE')
    monkeypatch.delenv('DCLR_BENCH_CHILD', raising=False)
    bench.main(['--gpus', '1', '--no-secondary'])
    assert calls[-1] == ('run', 1)
    # the plain single-GPU line (the driver's command) becomes the parent of the headline and secondary children
    monkeypatch.setattr(bench, 'run_with_secondary', lambda argv: calls.append(('secondary', list(argv))) or 0)
    with pytest.raises(SystemExit) as e:
        bench.main(['--gpus', '1', '--steps', '20', '--warmup', '5'])
    assert e.value.code == 0 and calls[-1] == ('secondary', ['--gpus', '1', '--steps', '20', '--warmup', '5'])
    monkeypatch.setenv('DCLR_BENCH_CHILD', '1')            # ... whose children run the measurement themselves
    bench.main(['--gpus', '1', '--steps', '20', '--warmup', '5'])
    assert calls[-1] == ('run', 1)


def test_secondary_block_is_attached_to_the_plain_single_gpu_line_only(bench, monkeypatch):
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    monkeypatch.delenv('DCLR_BENCH_CHILD', raising=False)
    want = lambda argv: bench.wants_secondary(bench.parse_args(argv), argv)          # noqa: E731
    assert want([]) and want(['--gpus', '1', '--steps', '20', '--warmup', '5']) and want(['--config', 'c2'])
    for argv in (['--strict'], ['--config', 'c4'], ['--clouds', 'ring'], ['--latency'], ['--no-secondary'], ['--h2d'],
                 ['--no-cpu-baseline'], ['--alone-only', '3'], ['--gpus', '2'], ['--same-batch'], ['--depth', '2']):
        assert not want(argv), argv
    monkeypatch.setenv('WORLD_SIZE', '1')                  # python -m torch.distributed.run --nproc-per-node 1
    assert not want([])
    monkeypatch.delenv('WORLD_SIZE')
    assert want([])
    monkeypatch.setenv('LD_PRELOAD', '/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so')   # under rocprofv3: the GPU is
    assert not want([])                                    # initialised in this process already -- no GPU children from it


def test_run_with_secondary_prints_one_line_with_every_pass_condensed(bench):
    """The headline child runs first with the caller's own arguments; then one child per secondary pass. A pass that fails
    fills its slot with the error; the headline line stands either way."""
    import io
    import types
    seen = []

    def fake_run(cmd, env=None, stdout=None, text=None, timeout=None):
        argv = cmd[2:]
        seen.append(argv)
        assert env['DCLR_BENCH_CHILD'] == '1'
        if '--strict' in argv:
            return types.SimpleNamespace(returncode=3, stdout='')
        if '--force-dist' in argv:
            line = {'value': 43000.0, 'unit': 'scan-pairs/s', 'ms_per_step': 0.186, 'steps': 200, 'warmup': 20,
                    'config': {'mode': 'grouped', 'workload': 'w'}, 'pose_delta_vs_oracle': 1e-6, 'roofline': None,
                    'ranks_seen': [0], 'collectives': {'all_gathers': 20, 'steps_per_all_gather': 10, 'bytes_per_rank': 2560},
                    'gather_check': {'checked': 20, 'ok': True}}
        elif '--h2d' in argv:
            line = {'value': 40000.0, 'unit': 'scan-pairs/s', 'ms_per_step': 0.2, 'steps': 200, 'warmup': 20,
                    'config': {'mode': 'grouped', 'workload': 'w'}, 'pose_delta_vs_oracle': 1e-6, 'roofline': None,
                    'h2d': {'bytes_per_step': 4194304.0, 'gb_per_s': 21.0, 'batches_per_copy': 2}}
        elif '--latency' in argv:
            line = {'value': 1234.0, 'unit': 'scan-pairs/s', 'ms_per_step': 0.81, 'steps': 20, 'warmup': 5,
                    'config': {'mode': 'latency', 'workload': 'one pair'}, 'pose_delta_vs_oracle': 1.5e-6,
                    'latency_ms': {'pairwise': {'median_ms': 0.81}, 'sequential': {'median_ms': 0.79}}, 'kernels_us': {'fps': 650.0},
                    'roofline': None}
        else:
            line = {'metric': 'scan-pairs/sec (2x16384 pts)', 'value': 40000.0 + len(seen), 'unit': 'scan-pairs/s', 'n_gpus': 1,
                    'steps': 20, 'warmup': 5, 'ms_per_step': 0.2, 'config': {'mode': 'grouped', 'workload': 'w'},
                    'pose_delta_vs_oracle': 1e-6, 'pose_delta_max': 2e-6, 'pose_delta_pairs': 4,
                    'roofline': {'kernel': 'head_conv_fused[80x1024]', 'frac': 0.2, 'frac_alone': 0.44, 'avg_us': 1000.0, 'alone_us': 460.0},
                    'roofline_sampler': {'kernel': 'fps_clouds[160x16384]', 'avg_us': 1400.0, 'alone_us': 700.0, 'samples_per_round': 3.1},
                    'latency_ms_per_batch': {'median': 9.6}}
        return types.SimpleNamespace(returncode=0, stdout='noise\n' + json.dumps(line) + '\n')

    out = io.StringIO()
    assert bench.run_with_secondary(['--gpus', '1', '--steps', '20', '--warmup', '5'], run=fake_run, out=out) == 0
    assert seen[0] == ['--gpus', '1', '--steps', '20', '--warmup', '5', '--no-secondary']      # the headline window, unchanged, first
    assert all('--no-secondary' in a and '--no-cpu-leg' in a for a in seen[1:]) and len(seen) == 1 + len(bench.SECONDARY)
    lines = out.getvalue().splitlines()
    assert len(lines) == 1
    doc = json.loads(lines[0])
    assert doc['value'] == 40001.0 and doc['steps'] == 20                                       # the headline's own figures
    sec = doc['secondary']
    assert set(sec) == {n for n, _ in bench.SECONDARY} | {'note'}
    # round 6: the reference API's own loop (`serial`), host-fed batches, the sequence workload, c5 on
    # ring scans and the one-rank RCCL pass are driver-observed too
    assert {'steady_200', 'strict', 'serial', 'ring', 'h2d', 'dist1', 'sequence', 'c4', 'c5', 'c5_ring', 'latency'} == set(sec) - {'note'}
    by_name = dict(bench.SECONDARY)
    assert '--no-overlap' in by_name['serial'] and '--force-dist' in by_name['dist1'] and '--sequence' in by_name['sequence']
    assert by_name['c5_ring'][:4] == ['--config', 'c5', '--clouds', 'ring'] and '--h2d' in by_name['h2d']
    assert sec['dist1']['ranks_seen'] == [0] and sec['dist1']['collectives']['all_gathers'] == 20 and sec['dist1']['gather_check']['ok']
    assert sec['h2d']['h2d']['gb_per_s'] == 21.0
    assert sec['strict']['error'].startswith('exit code 3') and '--strict' in sec['strict']['args']
    for name in ('steady_200', 'ring', 'c4', 'c5'):
        e = sec[name]
        assert e['value'] > 40000 and e['ms_per_step'] == 0.2 and e['pose_delta_max'] == 2e-6 and e['pose_delta_pairs'] == 4
        assert e['dominant_kernel'] == 'head_conv_fused[80x1024]' and e['frac'] == 0.2 and e['frac_alone'] == 0.44
        assert e['sampler']['alone_us'] == 700.0 and e['latency_ms_per_batch_median'] == 9.6
    assert sec['latency']['latency_ms'] == {'pairwise': 0.81, 'sequential': 0.79} and sec['latency']['pose_delta_max'] == 1.5e-6
    # the passes share ONE deadline -- once it is used up the remaining ones are skipped and the line still goes out
    now = [0.0]

    def slow_run(cmd, env=None, stdout=None, text=None, timeout=None):
        now[0] += 100.0                                                  # every child takes 100 s of the fake clock
        assert timeout is not None and timeout <= bench.HEADLINE_TIMEOUT_S
        return fake_run(cmd, env=env)
    out2 = io.StringIO()
    assert bench.run_with_secondary([], run=slow_run, out=out2, deadline_s=250, clock=lambda: now[0]) == 0
    sec2 = json.loads(out2.getvalue())['secondary']
    ran = [n for n, _ in bench.SECONDARY if 'skipped' not in str(sec2[n].get('error'))]
    assert ran == [n for n, _ in bench.SECONDARY][:3] and all('skipped' in sec2[n]['error'] for n, _ in bench.SECONDARY[3:])
    assert bench.HEADLINE_TIMEOUT_S + bench.SECONDARY_DEADLINE_S + 120 < 1500           # inside the driver's limit whatever hangs
    # the real child runner: its own process group, killed as a group on timeout
    import subprocess
    import sys
    with pytest.raises(subprocess.TimeoutExpired):
        bench._run_child([sys.executable, '-c', 'import time; time.sleep(30)'], dict(os.environ), 0.5)
    res = bench._run_child([sys.executable, '-c', 'print("{}")'], dict(os.environ), 30)
    assert res.returncode == 0 and res.stdout.strip() == '{}'
    # a failing headline is a failing run
    assert bench.run_with_secondary([], run=lambda *a, **k: types.SimpleNamespace(returncode=1, stdout=''), out=io.StringIO()) == 1


def test_output_gather_bookkeeping_world_size_2(bench, tmp_path):
    """The send / gather_every bookkeeping of bench.py's N > 1 path on gloo, two ranks (CPU)."""
    script = tmp_path / 'gather_worker.py'
    script.write_text(
        "import importlib.util, os, sys, torch, torch.distributed as dist\n"
        "spec = importlib.util.spec_from_file_location('b', sys.argv[1]); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)\n"
        "dist.init_process_group('gloo')\n"
        "rank, world = dist.get_rank(), dist.get_world_size()\n"
        "g = b.OutputGather(dist, world, 3, 4, 8, 'cpu')\n"
        "for step in range(7):\n"
        "    y = torch.full((4, 8), float(100 * rank + step))\n"
        "    if step % 2 == 0:\n"
        "        g.slot().copy_(y); g.put(None)\n"
        "    else:\n"
        "        g.put(y)\n"
        "    if step == 5:\n"
        "        got = g.gathered.view(world, 3, 4, 8)\n"
        "        assert g.collectives == 2 and g.filled == 0\n"
        "        for r in range(world):\n"
        "            for s in range(3):\n"
        "                assert bool((got[r, s] == 100 * r + 3 + s).all()), (r, s, got[r, s, 0, 0])\n"
        "g.flush(); assert g.collectives == 3 and g.filled == 0\n"
        "assert float(g.gathered.view(world, 3, 4, 8)[1, 0, 0, 0]) == 106.0\n"
        "seen = [None] * world; dist.all_gather_object(seen, rank); assert seen == [0, 1]\n"
        "dist.barrier(); dist.destroy_process_group()\n")
    envs = bench.rank_environments(2, bench.free_port())
    procs = [subprocess.Popen([sys.executable, str(script), os.path.join(ROOT, 'bench.py')], env=e) for e in envs]
    assert [p.wait(timeout=300) for p in procs] == [0, 0]


def test_traffic_figures_attach_only_to_the_kernels_they_were_measured_on(bench, tmp_path, monkeypatch):
    path = tmp_path / 'pmc_traffic.json'
    monkeypatch.setattr(bench, 'TRAFFIC_FILE', str(path))
    assert bench.load_traffic()[0] == {}
    path.write_text(json.dumps({'kernel_source_hash': 'deadbeef', 'commit': 'abc', 'configs': {'c2': {'x': {}}}}))
    got, why = bench.load_traffic()
    assert got == {} and 'other kernel sources' in why
    path.write_text(json.dumps({'kernel_source_hash': bench.kernel_source_hash(), 'commit': 'abc',
                                'configs': {'c2': {'head_conv_fused': {'bytes_per_launch': 123}}}}))
    got, why = bench.load_traffic()
    assert got['c2']['head_conv_fused']['bytes_per_launch'] == 123 and 'abc' in why


def test_traffic_is_keyed_by_the_launch_size_it_was_collected_at(bench):
    """profiles/collect.py stores the full span name beside each figure; a launch of another size gets nothing."""
    traffic = {'head_conv_fused': {'bytes_per_launch': 2.79e8, 'span': 'head_conv_fused[80x1024]', 'mfma_busy': 0.41},
               'knn_rows': {'bytes_per_launch': 2.7e7}}                                      # an old record without a span
    assert bench.traffic_for(traffic, 'head_conv_fused[80x1024]')['mfma_busy'] == 0.41
    assert bench.traffic_for(traffic, 'head_conv_fused[8x1024]') is None                     # --strict launch: not its figure
    assert bench.traffic_for(traffic, 'knn_rows[80x1024x20]') is None
    assert bench.traffic_for(traffic, 'fps_clouds[160x16384]') is None


def test_pose_check_covers_every_batch_before_any_batch_gets_a_second_pair(bench, monkeypatch):
    calls = []

    def fake(y, x, cfg, sd, pairs_cfg, sequence, rows=None):
        calls.append((int(x[0]), tuple(rows)))
        return [1e-6]
    monkeypatch.setattr(bench, 'pose_deltas', fake)
    import torch
    recent = [(torch.tensor([i]), torch.zeros(8, 8)) for i in range(5)]
    deltas, available, covered = bench.pose_check(recent, None, None, 8, False, budget_s=0.0)   # no budget: one pair per batch
    assert available == 40 and covered == 5 and len(deltas) == 5
    assert calls == [(i, (0,)) for i in range(5)]
    calls.clear()
    deltas, _, covered = bench.pose_check(recent, None, None, 8, False, budget_s=1e9)           # ample budget: every pair
    assert len(deltas) == 40 and covered == 5 and calls[5] == (0, (1,))
    calls.clear()
    deltas, _, covered = bench.pose_check(recent, None, None, 8, False, budget_s=1e9, max_pairs=3)   # the secondary passes' cap
    assert len(deltas) == 3 and [c[0] for c in calls] == [0, 2, 4]                               # spread over the window
    calls.clear()
    deltas, _, _ = bench.pose_check(recent, None, None, 8, False, budget_s=1e9, max_pairs=7)
    assert len(deltas) == 7 and calls[5] == (0, (1,))


def test_cpu_baseline_reports_batch_one_and_the_configurations_batch(bench):
    from deepclr_amd import synthetic
    cfg = synthetic.model_cfg('kitti')
    out = bench.cpu_baseline(cfg, synthetic.random_state_dict(cfg, seed=0), 'kitti', 256, budget_s=0.2, pairs_cfg=8)
    assert out['kind'] == 'port' and out['value'] > 0 and out['cores'] >= 1 and 'batch 1' in out['sample']
    assert out['batched']['batch'] == 8 and out['batched']['value'] > 0 and 'calls of 8 pairs' in out['batched']['sample']
    one = bench.cpu_baseline(cfg, synthetic.random_state_dict(cfg, seed=0), 'kitti', 256, budget_s=0.1, pairs_cfg=1)
    assert 'batched' not in one


def test_new_arguments_parse(bench):
    a = bench.parse_args(['--same-batch', '--pose-budget', '5'])
    assert a.same_batch and a.pose_budget == 5.0
    assert not bench.parse_args([]).same_batch


def test_ranks_get_disjoint_core_slices(bench):
    """Eight ranks on one host: contiguous, disjoint slices of the cores the process may use; no pinning when the host has
    fewer cores than ranks or for a single rank."""
    avail = list(range(4, 68))                                                 # 64 cores, not starting at 0
    slices = [bench.rank_cores(r, 8, avail) for r in range(8)]
    assert all(len(sl) == 8 for sl in slices) and slices[0] == list(range(4, 12)) and slices[7] == list(range(60, 68))
    assert len(set(c for sl in slices for c in sl)) == 64
    assert bench.rank_cores(0, 1, avail) is None and bench.rank_cores(3, 8, [0, 1, 2]) is None
    assert bench.rank_cores(1, 3, list(range(8))) == [2, 3]                     # 8 cores over 3 ranks: 2 each, 2 left unassigned


def test_eight_rank_launch_with_the_stub_gathers_on_every_rank(bench):
    """`python bench.py --gpus 8 --cpu-stub`: the self-launcher's eight children over gloo, one all-gather per 4 steps,
    the gathered buffer verified on EVERY rank (MIN all-reduce of the per-rank verdicts), ranks pinned to disjoint cores
    where the host has enough of them."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--cpu-stub', '--steps', '12',
                          '--warmup', '4', '--gather-every', '4'], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line['ranks_seen'] == list(range(8)) and line['n_gpus'] == 8 and line['gather_check'] is True
    assert line['collectives']['all_gathers'] == 4 and line['config']['parallelism'] == 'dp8'     # 1 in the warm-up + 3
    if len(os.sched_getaffinity(0)) >= 8:
        assert line['host_cores_of_rank0'] is not None
