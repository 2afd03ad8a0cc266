"""bench.py without a GPU: the algorithmic work it prices kernels with, and the fields of its JSON contract."""
import importlib.util
import os

import pytest

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
    pairs, n = 8, 16384
    bound, flops = bench.algorithmic_work('flow_embedding', cfg, pairs, n, 2 * pairs)
    rows = pairs * 1024 * 20
    assert bound == 'mfma' and flops == 2.0 * rows * (128 * 128 + 128 * 256) + 2.0 * rows * 128 * 5
    assert abs(flops / pairs - 2.01e9) < 0.05e9                       # 2.01 GFLOP per pair (+ layer-1 rest)
    bound, flops = bench.algorithmic_work('head_conv_fused', cfg, pairs, n, 2 * pairs)
    assert bound == 'mfma' and abs(flops / pairs - 2.15e9) < 0.02e9    # 1024 x 1,049,344 MAC (+ 5 padded input columns)
    bound, flops = bench.algorithmic_work('linear_pair[2x8192x128x64]', cfg, pairs, n, 2 * pairs)
    assert bound == 'mfma' and flops == 4.0 * 8192 * 128 * 64
    bound, nbytes = bench.algorithmic_work('fps_clouds', cfg, pairs, n, 2 * pairs)
    assert bound == 'hbm' and nbytes == 16 * (16384 * 4 * 4 + 1024 * 4)
    bound, nbytes = bench.algorithmic_work('sa_msg_fused', cfg, pairs, n, 2 * pairs)
    assert bound == 'hbm' and nbytes == 16 * (16384 * 16 + 1024 * 4 + 1024 * 68 * 4)


def test_peaks_and_workload_constants(bench):
    assert bench.PAIRS_PER_GPU == 8 and bench.POINTS == 16384                  # BASELINE.json configs[1]
    assert bench.FP32_MATRIX_PEAK_TFLOPS == 157.3 and bench.HBM_PEAK_GBS == 8000.0
    assert abs(bench.F16_MATRIX_PEAK_TFLOPS / bench.SPLIT_PRODUCTS - 838.9) < 0.1
    assert os.environ.get('GPU_MAX_HW_QUEUES') is not None                     # set on import, before HIP initialises


def test_launch_timer_sampling_is_coprime_with_group_sizes(bench):
    for group in (2, 3, 4):
        assert bench.LaunchTimer.SAMPLE_EVERY % group != 0
