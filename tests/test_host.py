"""CPU tests of the host layer: config, labels, state_dict layout, C-ABI surface, failure behaviour."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch
import yaml

from deepclr_amd import lib, synthetic
from deepclr_amd.config import load_model_config, model_config_from_dict
from deepclr_amd.labels import LabelType
from deepclr_amd.models import build_model, load_trained_model, ModelInferenceHelper, ModelType

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    """include/deepclr_amd.h <-> libdeepclr_amd.so <-> lib.SIGNATURES agree (no compute call: no GPU here)."""
    header = open(os.path.join(ROOT, 'include', 'deepclr_amd.h')).read()
    declared = set(re.findall(r'\b(dclr_[a-z0-9_]+)\s*\(', header)) - {'dclr_stream_t'}
    assert declared == set(lib.SIGNATURES), declared ^ set(lib.SIGNATURES)
    assert os.path.exists(lib.LIB_PATH), 'run python -m deepclr_amd.build'
    handle = ctypes.CDLL(lib.LIB_PATH)
    for name in declared:
        assert hasattr(handle, name), name
    assert lib.load().dclr_version() >= 1
    assert lib.load().dclr_error_string(-1).decode().startswith('invalid argument')
    assert lib.load().dclr_error_string(-2).decode().startswith('configuration not supported')


def test_state_dict_layout_matches_reference():
    """Key names / shapes of SURVEY.md section 8a-11 (verified against the reference's own modules by
    tests/golden/make_golden.py, which loads the same synthetic state_dict with strict=True)."""
    for kind, n_params in (('kitti', 1_778_312), ('modelnet', 1_778_280)):
        cfg = synthetic.model_cfg(kind)
        model = build_model(model_config_from_dict(cfg))
        sd = model.state_dict()
        shapes = synthetic.state_dict_shapes(cfg)
        assert list(sd.keys()) == list(shapes.keys())
        assert all(tuple(sd[k].shape) == shapes[k] for k in shapes)
        assert sum(v.numel() for v in sd.values()) == n_params
        out_bias = sd['_merge_layers.1.output.bias']
        assert out_bias.tolist() == [1.0, 0, 0, 0, 0, 0, 0, 0]           # LabelType.bias for dual quaternions
        assert float(sd['_merge_layers.1.conv._sequential.0._sequential.0.bias'].abs().max()) == 0.0
    assert model.get_input_dim() == 3 and not model.has_loss() and model.get_loss_weights() == {}
    # a `transform` module (reference deepclr.py:447,453-464): `_cloud_layers` = [transform, cloud features], keys shifted by
    # one, the feature module fed with the transform's 3 + 64 channels; such a model runs module by module
    from helpers import small_transform_cfg
    tr = build_model(model_config_from_dict(small_transform_cfg()))
    shapes = synthetic.state_dict_shapes(small_transform_cfg())
    assert {k: tuple(v.shape) for k, v in tr.state_dict().items()} == dict(shapes)
    assert shapes['_cloud_layers.0._sa0.mlps.0.layer0.conv.weight'] == (16, 4, 1, 1)
    assert shapes['_cloud_layers.1._sa0.mlps.1.layer0.conv.weight'] == (48, 67, 1, 1)
    assert len(tr._cloud_layers) == 2 and not tr._rows_path and tr.npoint == 64


def test_load_model_config_and_trained_model(tmp_path):
    cfg_d = synthetic.model_cfg('kitti')
    cfg_d['params']['loss'] = {'name': 'TransformLoss', 'params': {'p': 2, 'sx': 1, 'sq': 1}}   # as tests/model/deepclr.yaml
    path = tmp_path / 'model_config.yaml'
    path.write_text(yaml.safe_dump(cfg_d))
    sd = synthetic.random_state_dict(cfg_d, 1)
    weights = tmp_path / 'weights.tar'
    torch.save(sd, weights)
    cfg = load_model_config(str(path), str(weights))
    assert cfg.label_type is LabelType.POSE3D_DUAL_QUAT and cfg.model_type is ModelType.DEEPCLR
    assert cfg.params.cloud_features.name == 'SetAbstraction' and cfg.weights == str(weights)
    model = load_trained_model(cfg)
    assert model.has_loss()
    for k, v in sd.items():
        assert torch.equal(model.state_dict()[k], v)
    with pytest.raises(RuntimeError):
        model_config_from_dict({**synthetic.model_cfg('kitti'), 'point_dim': 5})


def test_unsupported_configurations_fail_loudly():
    base = synthetic.model_cfg('kitti')
    for mutate, err in ((lambda c: c['params']['merge']['params'].update(k=-1), ValueError),
                        (lambda c: c.update(point_dim=2, input_dim=3), (NotImplementedError, RuntimeError, AssertionError))):
        cfg = synthetic.model_cfg('kitti')
        mutate(cfg)
        with pytest.raises(err):
            build_model(model_config_from_dict(cfg))
    cfg = synthetic.model_cfg('kitti')
    cfg['params']['merge']['params'].update(k=100)                # any k builds since round 5 (composed path beyond 32)
    assert not build_model(model_config_from_dict(cfg))._rows_path
    assert base == synthetic.model_cfg('kitti')
    cfg = synthetic.model_cfg('kitti')
    cfg['params']['merge']['params'].update(k=0)                  # GlobalGrouping (deepclr.py:186-187) is supported
    assert build_model(model_config_from_dict(cfg)) is not None
    cfg = synthetic.model_cfg('kitti')
    cfg['params'].update(batch_norm=True, dropout=0.5)            # norm layers and dropout build since round 5 (eval: folded / identity)
    assert build_model(model_config_from_dict(cfg)) is not None


def test_configurations_beyond_the_fused_shapes_build_and_take_the_composed_path():
    """Other layer widths / k up to 64 / more input features / append_features = False (reference deepclr.py:50-70,
    180-199 accept any): the model builds with the reference's state_dict layout and marks itself for the module-by-module
    path; its row-pipeline entry points say so instead of computing something else."""
    from helpers import custom_features_cfg, custom_widths_cfg
    assert build_model(model_config_from_dict(synthetic.model_cfg('kitti')))._rows_path
    for factory, keys in ((custom_widths_cfg, {'_merge_layers.0._embedding._conv._sequential.0._sequential.0.weight': (64, 131, 1),
                                               '_cloud_layers.0._sa0.mlps.0.layer2.conv.weight': (64, 32, 1, 1)}),
                          (custom_features_cfg, {'_merge_layers.0._embedding._conv._sequential.0._sequential.0.weight': (96, 83, 1),
                                                 '_cloud_layers.0._sa0.mlps.1.layer0.conv.weight': (16, 6, 1, 1),
                                                 '_merge_layers.1.conv._sequential.0._sequential.0.weight': (64, 67, 1)})):
        cfg = factory()
        model = build_model(model_config_from_dict(cfg))
        sd = synthetic.random_state_dict(cfg, seed=2)
        for key, shape in keys.items():
            assert tuple(sd[key].shape) == shape
        model.load_state_dict(sd, strict=True)
        assert not model._rows_path
        with pytest.raises(NotImplementedError, match='composed'):
            model._cloud_layers[0].forward_rows(torch.zeros(2, 64, cfg['input_dim']))
        with pytest.raises(NotImplementedError, match='composed'):
            model._merge_layers[0].forward_rows(torch.zeros(128, 68), 1, 64)
    for mutate in (lambda c: c['params']['merge']['params'].update(k=40),
                   lambda c: c['params']['merge']['params'].update(mlp=[64, 64, 128]),
                   lambda c: c['params']['cloud_features']['params'].update(mlps=[[[32, 32, 64], [16, 16, 32]]])):
        cfg = synthetic.model_cfg('kitti')
        mutate(cfg)
        assert not build_model(model_config_from_dict(cfg))._rows_path


def test_second_set_abstraction_level_builds_with_the_reference_key_layout():
    """deepclr.py:72-83: a second level `_sa1` whose mlp specs start with their input width; its parameters sit under
    `_cloud_layers.0._sa1.mlps.{scale}.layer{j}.conv.*` and load strictly."""
    from helpers import small_two_level_cfg
    cfg = small_two_level_cfg()
    model = build_model(model_config_from_dict(cfg))
    sd = synthetic.random_state_dict(cfg, seed=1)
    assert sd['_cloud_layers.0._sa1.mlps.1.layer0.conv.weight'].shape == (48, 67, 1, 1)
    model.load_state_dict(sd, strict=True)
    sa = model._cloud_layers[0]
    assert sa._sa0.fused and not sa._sa1.fused and sa.npoint == 64 and sa.output_dim() == 67


def test_no_cpu_fallback():
    model = build_model(model_config_from_dict(synthetic.model_cfg('kitti')))
    with pytest.raises(RuntimeError):
        model(torch.zeros(2, 64, 4))
    helper = ModelInferenceHelper(model)
    with pytest.raises(RuntimeError):
        helper.predict(torch.zeros(64, 3), torch.zeros(64, 4))          # too few columns
    with pytest.raises(RuntimeError):
        helper.predict(torch.zeros(64, 4))                              # template missing


def test_stack_subsamples_larger_cloud():
    a, b = torch.rand(10, 4), torch.rand(7, 4)
    s = ModelInferenceHelper.stack(a, b)
    assert s.shape == (2, 7, 4) and torch.equal(s[1], b)
    rows = {tuple(r.tolist()) for r in a}
    assert all(tuple(r.tolist()) in rows for r in s[0])


def test_label_types():
    lt = LabelType.create('pose3d_dual_quat')
    assert lt.dim == 8 and len(lt.names) == 8 and lt.bias[0] == 1.0
    ident = lt.to_matrix(np.array([1, 0, 0, 0, 0, 0, 0, 0], dtype=np.float32))
    assert np.abs(ident - np.eye(4)).max() < 1e-7
    q = LabelType.POSE3D_QUAT
    m = q.to_matrix(np.array([1.0, 2.0, 3.0, 0.5, 0.5, 0.5, 0.5]))
    assert np.allclose(m[:3, 3], [1, 2, 3]) and np.allclose(m[:3, :3] @ m[:3, :3].T, np.eye(3))
    assert np.allclose(q.to_matrix(q.from_matrix(m)), m)
    # POSE3D_EULER (reference labels.py:54-58,82-86; transforms3d absent: parity unpinned -- checked by recomposition, by the
    # defining rotation order Rz(yaw) Ry(pitch) Rx(roll) on single-axis cases, and against the dual-quaternion branch)
    e = LabelType.POSE3D_EULER
    assert e.dim == 6 and e.bias is None and e.names[3:] == ['roll', 'pitch', 'yaw']
    assert np.allclose(e.to_matrix(np.zeros(6)), np.eye(4))
    rz = e.to_matrix(np.array([0, 0, 0, 0, 0, 90.0]))[:3, :3]            # degrees; yaw about z: x -> y
    assert np.allclose(rz @ [1, 0, 0], [0, 1, 0]) and np.allclose(rz @ [0, 0, 1], [0, 0, 1])
    rx = e.to_matrix(np.array([0, 0, 0, 90.0, 0, 0]))[:3, :3]            # roll about x: y -> z
    assert np.allclose(rx @ [0, 1, 0], [0, 0, 1])
    both = e.to_matrix(np.array([0, 0, 0, 90.0, 0, 90.0]))[:3, :3]       # static axes: roll first, then yaw
    assert np.allclose(both, rz @ rx)
    rng = np.random.default_rng(5)
    for _ in range(20):
        lab = np.concatenate((rng.normal(size=3), rng.uniform(-89, 89, size=3)))
        m = e.to_matrix(lab.copy())
        assert np.allclose(m[:3, :3] @ m[:3, :3].T, np.eye(3)) and np.isclose(np.linalg.det(m[:3, :3]), 1.0)
        assert np.allclose(e.from_matrix(m), lab, atol=1e-9)             # inside the principal range the pair inverts
        assert np.allclose(lt.to_matrix(lt.from_matrix(m)), m, atol=1e-12)        # the same pose through the dual-quaternion branch
        assert np.allclose(e.to_matrix(e.from_matrix(m, scale=2.0), scale=2.0), m)
    sing = e.to_matrix(np.array([1.0, 2.0, 3.0, 20.0, 90.0, 30.0]))      # pitch = 90 deg: yaw reads 0, roll takes the rest
    back = e.from_matrix(sing)
    assert back[5] == 0.0 and np.allclose(e.to_matrix(back), sing, atol=1e-12)


def test_synthetic_inputs_are_deterministic():
    a = synthetic.make_batch('kitti', 2, 128)
    b = synthetic.make_batch('kitti', 2, 128)
    assert a.shape == (4, 128, 4) and a.dtype == np.float32 and np.array_equal(a, b)
    assert not np.array_equal(a[0], synthetic.make_batch('kitti', 1, 128, first_pair=1)[0])
    m = synthetic.make_batch('modelnet', 1, 64)
    assert m.shape == (2, 64, 3) and np.abs(m).max() < 1.5


def test_ring_scan_clouds_have_lidar_density():
    """bench.py --clouds ring: 64 rings x azimuth, every second point kept, exactly n points in scan order; near the sensor a
    1 m ball holds hundreds of points (the Gaussian clouds: ~20), so ball queries reach the nsample caps 512 / 1024."""
    from scipy.spatial import cKDTree
    r = synthetic.make_batch('ring', 1, 16384)
    assert r.shape == (2, 16384, 4) and r.dtype == np.float32 and np.isfinite(r).all()
    assert np.array_equal(r, synthetic.make_batch('ring', 1, 16384))
    t = r[0]
    rng_xy = np.linalg.norm(t[:, :2], axis=1)
    assert rng_xy.min() > 1.0 and np.linalg.norm(t[:, :3], axis=1).max() < 81.0 and 0.0 <= t[:, 3].min() and t[:, 3].max() <= 1.0
    counts = cKDTree(t[:, :3]).query_ball_point(t[::8, :3], 1.0, return_length=True)
    g = synthetic.make_batch('kitti', 1, 16384)[0]
    counts_g = cKDTree(g[:, :3]).query_ball_point(g[::8, :3], 1.0, return_length=True)
    assert counts.max() >= 1024 and counts.mean() > 10 * counts_g.mean() and counts_g.max() < 512
    big = synthetic.make_batch('ring', 1, 65536)[0]
    assert big.shape == (65536, 4)


def test_reference_import_names_resolve_to_this_package(tmp_path):
    """What scripts/inference.py:9-13 and scripts/timing.py:6-10 import must exist under the reference's names."""
    import deepclr_amd.models
    from deepclr.config import load_model_config, load_config, Config, Mode                      # noqa: F401
    from deepclr.data import create_input_dataflow, make_data_loader, LabelType                  # noqa: F401
    from deepclr.evaluation import load_scenario, Evaluator
    from deepclr.models import load_trained_model, build_model, ModelInferenceHelper              # noqa: F401
    from deepclr.utils.logging import create_logger
    from deepclr.utils.tensor import prepare_tensor
    assert ModelInferenceHelper is deepclr_amd.models.ModelInferenceHelper
    with pytest.raises(RuntimeError):
        create_input_dataflow('kitti_odometry_velodyne', 'x.lmdb', shuffle=False)
    assert prepare_tensor(torch.ones(2), device='cpu').device.type == 'cpu'
    log = create_logger('test_host_logger')
    assert log is create_logger('test_host_logger') and len(log.handlers) <= 1     # never attached twice
    scen = tmp_path / 'scen.yaml'
    scen.write_text("name: kitti_07\ndataset_type: kitti_odometry_velodyne\nsequential: True\n"
                    "data:\n  '07': '${HOME}/odometry/07.lmdb'\n")
    cfg = load_scenario(str(scen), with_method=False)
    assert cfg.sequential is True and cfg.dataset_type.name == 'KITTI_ODOMETRY_VELODYNE'
    assert '$' not in cfg.data['07'] and cfg.data['07'].endswith('odometry/07.lmdb')
    eval_cfg = cfg.copy()                                         # scripts/inference.py:64-70
    eval_cfg.method.name = 'DEEPCLR'
    eval_cfg.method.params.model_name = 'kitti_00-06'
    eval_cfg.write_file(str(tmp_path / 'scenario.yaml'), invalid=True, internal=True)
    assert 'model_name: kitti_00-06' in (tmp_path / 'scenario.yaml').read_text()
    assert cfg.method.name is None
    with pytest.raises(RuntimeError):
        load_scenario(str(scen), with_method=True)
    assert isinstance(Evaluator(), deepclr_amd.evaluation.Evaluator)


def test_load_config_and_tensor_backed_loader_for_timing_script(tmp_path):
    """scripts/timing.py:57 `load_config(args.config, Mode.TEST)` and :23 `make_data_loader(cfg, is_train=False,
    batch_size=1)`: `extends:` chains, finalized model section, device, and batches in the reference's layout
    (data/build.py:62-98). Dataset types that need the LMDB / dataflow readers raise."""
    from deepclr.config import load_config, Mode
    from deepclr.data import make_data_loader
    from deepclr.models import build_model
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = load_config(os.path.join(root, 'configs', 'timing_synthetic_kitti.yaml'), Mode.TEST)
    assert cfg.device == 'cuda' and cfg.mode == Mode.TEST and cfg.extends is None
    assert cfg.model.label_type.name == 'POSE3D_DUAL_QUAT' and cfg.model.model_type.name == 'DEEPCLR'
    assert cfg.model.params.cloud_features.params.npoint == [1024] and cfg.model.params.merge.params.k == 20
    model = build_model(cfg.model)                                       # timing.py:15
    assert sum(p.numel() for p in model.parameters()) == 1778312
    loader = make_data_loader(cfg, is_train=False, batch_size=1)         # timing.py:23
    batches = list(loader)
    assert len(batches) == len(loader) == 32
    b = batches[3]
    assert tuple(b['x'].shape) == (2, 16384, 4) and b['x'].dtype == torch.float32
    assert tuple(b['y'].shape) == (1, 8) and tuple(b['m'].shape) == (2, 4, 4) and tuple(b['t'].shape) == (1, 2)
    np.testing.assert_allclose(cfg.model.label_type.to_matrix(b['y'][0].numpy().copy()),
                               synthetic.kitti_like_pair(100003, 16384)[2], atol=1e-6)
    again = list(make_data_loader(cfg, is_train=False, batch_size=1))
    assert torch.equal(b['x'], again[3]['x'])                             # seeded: the same pairs every time
    big = make_data_loader(cfg, is_train=True, batch_size=5)
    first = next(iter(big))
    assert tuple(first['x'].shape) == (10, 16384, 4) and len(big) == 7
    # a child file overrides its parent section by section; environment variables in paths are expanded
    child = tmp_path / 'child.yaml'
    child.write_text("extends: '%s'\ndevice: cuda:1\nbase_dir: '${HOME}/runs'\n"
                     "data:\n  dataset_type: kitti_odometry_velodyne\n  validation: ['${HOME}/odometry/07.lmdb']\n"
                     "model:\n  params:\n    merge:\n      params:\n        k: 16\n"
                     % os.path.join(root, 'configs', 'timing_synthetic_kitti.yaml'))
    cfg2 = load_config(str(child), Mode.TEST)
    assert cfg2.device == 'cuda:1' and '$' not in cfg2.base_dir and '$' not in cfg2.data.validation[0]
    assert cfg2.model.params.merge.params.k == 16 and cfg2.model.params.merge.params.radius == 10.0
    assert cfg2.data.points == 16384                                      # inherited
    with pytest.raises(RuntimeError, match='LMDB'):
        make_data_loader(cfg2, is_train=False, batch_size=1)
    with pytest.raises(RuntimeError, match='checkpoint'):
        load_config(str(child), Mode.CONTINUE)
    nomodel = tmp_path / 'nomodel.yaml'
    nomodel.write_text("device: cuda\n")
    with pytest.raises(RuntimeError, match='missing required parameters'):
        load_config(str(nomodel), Mode.TEST)
    ref = '/root/reference/configs/training/kitti_00-06.yaml'              # the reference's own files, where present
    if os.path.exists(ref):
        cfg3 = load_config(ref, Mode.TEST)
        assert cfg3.identifier == 'kitti_00-06' and cfg3.model.input_dim == 4 and cfg3.data_loader.batch_size == 5
        assert cfg3.model.params.output.params.mlp == [256, 256, 512, 512, 1024]


def test_entry_points_reject_bad_arguments_before_touching_the_gpu():
    """Argument validation comes first in every entry point: null pointers, sizes that break a stated relation and
    unsupported shapes return DCLR_E_INVALID / DCLR_E_UNSUPPORTED (never a launch), so this runs without a GPU."""
    handle = lib.load()
    inval, unsup = -1, -2
    assert handle.dclr_furthest_point_sampling(1, 8, 4, None, None, None, None) == inval
    assert handle.dclr_ball_query(1, 8, 4, 0.5, 4, None, None, None, None) == inval
    assert handle.dclr_knn(1, 8, 8, 4, None, None, None, None, None) == inval
    assert handle.dclr_merge_forward(None, None, None) == inval
    args = lib.MergeArgs()                                   # all zero: sizes invalid
    assert handle.dclr_merge_forward(ctypes.byref(args), None, None) == inval
    # ABI 0.2: both argument structs open with their own size, and a caller built against another header
    # (0.1 had neither that member nor `overflow`) is rejected instead of being read past its end
    assert handle.dclr_version() == 2
    assert args.struct_size == ctypes.sizeof(lib.MergeArgs) and lib.CloudArgs().struct_size == ctypes.sizeof(lib.CloudArgs)
    cloud = lib.CloudArgs()
    cloud.b, cloud.n, cloud.c, cloud.npoint, cloud.n_scales = 2, 64, 4, 16, 1
    cloud.struct_size -= 8
    assert handle.dclr_cloud_forward(ctypes.byref(cloud), None, None, None) == inval
    assert [handle.dclr_flow_f16_tile(k) for k in (20, 28, 29, 30, 32)] == [16, 16, 32, 32, 32]
    assert handle.dclr_prepare_cloud(10, 4, None, 1, 0, 0.0, 1.0, 4, None, None, None, None) == inval
    assert handle.dclr_prepare_cloud_blocks(10, 2, 2) == inval          # start must be < nth
    assert handle.dclr_prepare_cloud_blocks(4097, 2, 1) == 2 and handle.dclr_prepare_cloud_blocks(4099, 2, 1) == 3
    assert handle.dclr_fps_workspace_bytes(2, 16384) == 0 and handle.dclr_fps_workspace_bytes(0, 100) == inval
    assert handle.dclr_fps_workspace_bytes(2, 65536) == 2 * 65536 * 26
    assert handle.dclr_fps_workspace_bytes(1, 20000) == 32768 * 26
    n_groups, size = ctypes.c_int(0), ctypes.c_int(0)
    assert handle.dclr_fps_group_layout(16384, ctypes.addressof(n_groups), ctypes.addressof(size)) == 0
    assert (n_groups.value, size.value) == (64, 256)
    assert handle.dclr_fps_group_layout(70000, ctypes.addressof(n_groups), ctypes.addressof(size)) == unsup
    assert b'invalid argument' in handle.dclr_error_string(inval)
    assert b'not supported' in handle.dclr_error_string(unsup)


def test_loss_values_match_reference_loss_modules():
    """tests/golden/losses.npz holds values computed by the reference's TransformLoss / TransformUncertaintyLoss /
    AccumulatedLoss on random dual-quaternion labels (make_golden.py); CPU tensors: the losses are plain torch."""
    from deepclr_amd.models.deepclr import AccumulatedLoss, TransformLoss, TransformUncertaintyLoss
    from deepclr_amd import losses
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'losses.npz'))
    y_pred, y = torch.from_numpy(g['y_pred']), torch.from_numpy(g['y'])
    lt = LabelType.POSE3D_DUAL_QUAT
    for p in (1, 2):
        t, r = losses.transform_losses(y_pred, y, lt, p)
        fixed = TransformLoss(lt, p=p, sx=1.5, sq=40.0)
        learned = TransformUncertaintyLoss(lt, p=p, sx=0.3, sq=-2.5)
        both = AccumulatedLoss([fixed, TransformLoss(lt, p=p, sx=0.5, sq=2.0)])
        got = np.array([t.item(), r.item(), fixed(y_pred, y).item(), learned(y_pred, y).item(), both(y_pred, y).item()])
        np.testing.assert_allclose(got, g['p%d' % p], rtol=1e-6)
        assert set(learned.get_weights()) == {'sx', 'sq'} and both.get_weights() == {}
    with pytest.raises(RuntimeError):
        losses.transform_losses(y_pred * float('nan'), y, lt, 2)


def test_hot_kernels_do_not_spill():
    """The build records every kernel's registers / scratch (deepclr_amd/build.py): nothing on the hot path may use
    scratch memory -- the 16384-point sampler once picked up 196 bytes of it from an innocent-looking epilogue and ran
    20 % slower. Allowed: the A/B and fallback kernels that the default path never launches."""
    from deepclr_amd import build
    usage = build.kernel_usage()
    assert len(usage) >= 60, 'run python -m deepclr_amd.build'
    # (round 6: the sampler's fallback kernels fps_stream_kernel<32> / <64> -- no workspace, or hipMallocAsync refused --
    # no longer spill either: 252 / 904 bytes before)
    single_sample_ab = re.compile(r'fps_paged_kernelILi\d+ELi0E')     # DCLR_FPS_SINGLE=1
    spilled = {k: v['scratch'] for k, v in usage.items() if v['scratch'] and not single_sample_ab.search(k)}
    assert not spilled, spilled
    sampler = [v for k, v in usage.items() if 'fps_pruned_kernelILi1024ELi16ELi4ELi3E' in k]      # the table mode (default)
    assert sampler and sampler[0]['vgprs'] <= 128 and sampler[0]['occupancy'] >= 4      # 16 waves = one cloud per CU


def test_array_backed_input_dataflow_yields_the_reference_structure(tmp_path):
    """create_input_dataflow over .npz files: the unified data-point structure of the reference
    (data/datasets/build.py:97-130), pairs of consecutive frames with transform = inv(pose_i) pose_{i+1}."""
    from deepclr.data import create_input_dataflow
    from deepclr.data import DatasetType
    rng = np.random.default_rng(0)
    clouds = rng.normal(size=(4, 50, 4))                                   # float64 on disk -> float32 out (ToFloat32)
    poses = np.tile(np.eye(4), (4, 1, 1))
    for i in range(4):
        poses[i, 0, 3] = 1.5 * i
        poses[i, :2, :2] = [[np.cos(0.1 * i), -np.sin(0.1 * i)], [np.sin(0.1 * i), np.cos(0.1 * i)]]
    seq = tmp_path / 'seq_07.npz'
    np.savez(seq, clouds=clouds, poses=poses, timestamps=np.array([0.0, 0.1, 0.2, 0.3]))
    df = create_input_dataflow(DatasetType.KITTI_ODOMETRY_VELODYNE, str(seq), shuffle=False)
    df.reset_state()
    items = list(df)
    assert len(df) == 3 and len(items) == 3
    for i, d in enumerate(items):
        assert sorted(d) == ['augmentations', 'clouds', 'dataset', 'idx', 'timestamps', 'transform']
        assert d['dataset'] == 'seq_07' and d['idx'] == [i, i + 1] and d['augmentations'] == [None, None]
        assert d['timestamps'] == [pytest.approx(0.1 * i), pytest.approx(0.1 * (i + 1))]
        assert d['clouds'][0].dtype == np.float32 and np.array_equal(d['clouds'][1], clouds[i + 1].astype(np.float32))
        np.testing.assert_allclose(d['transform'], np.linalg.inv(poses[i]).dot(poses[i + 1]), atol=1e-6)
    pairs = tmp_path / 'pairs.npz'
    np.savez(pairs, templates=clouds[:2], sources=clouds[2:], transforms=poses[:2])
    items = list(create_input_dataflow(DatasetType.GENERIC, str(pairs)))
    assert len(items) == 2 and np.array_equal(items[1]['clouds'][1], clouds[3].astype(np.float32))
    objs = tmp_path / 'objects.npz'
    np.savez(objs, clouds=clouds[:, :, :3])
    items = list(create_input_dataflow(DatasetType.MODELNET40, str(objs)))
    assert len(items) == 4 and np.array_equal(items[2]['clouds'][0], items[2]['clouds'][1]) \
        and items[2]['clouds'][0] is not items[2]['clouds'][1] and np.array_equal(items[2]['transform'], np.eye(4))
    with pytest.raises(RuntimeError, match='LMDB'):
        create_input_dataflow(DatasetType.GENERIC, str(tmp_path / 'data.lmdb'))


def test_flat_parameters_follows_late_registrations_and_replicas():
    """The cached (owner dict, name) slots behind flat_parameters: a parameter or submodule registered after the first
    call is picked up (the packed-weight caches key on this list), replaced parameters are seen, and a shallow copy of
    the module does not reuse the original's list."""
    import copy
    from torch import nn
    from deepclr_amd.models.helper import flat_parameters
    net = torch.nn.Sequential(torch.nn.Linear(3, 4), torch.nn.ReLU())
    first = flat_parameters(net)
    assert [id(p) for p in first] == [id(p) for p in net.parameters()]
    net[0].extra = torch.nn.Parameter(torch.zeros(2))                      # registered after the first call
    assert [id(p) for p in flat_parameters(net)] == [id(p) for p in net.parameters()] and len(flat_parameters(net)) == 3
    net.add_module('tail', torch.nn.Linear(4, 1))                          # a submodule added later
    assert [id(p) for p in flat_parameters(net)] == [id(p) for p in net.parameters()] and len(flat_parameters(net)) == 5
    net[0].weight = torch.nn.Parameter(torch.ones(4, 3))                   # replaced in place: same slot, new tensor
    assert flat_parameters(net)[0] is net[0].weight
    twin = copy.copy(net)                                                  # __dict__ copied shallowly, as DataParallel replicas are
    twin._modules = dict(net._modules)
    twin._modules['0'] = torch.nn.Linear(3, 4)
    assert flat_parameters(twin)[0] is twin._modules['0'].weight and flat_parameters(net)[0] is net[0].weight
    shared = torch.nn.Linear(2, 2)
    tied = torch.nn.Sequential(shared, shared)                             # the same parameters twice: listed once, as parameters() does
    assert len(flat_parameters(tied)) == 2
    # removing a submodule fires no registration hook: del / pop / delattr are seen all the same
    seq = nn.Sequential(nn.Linear(2, 2), nn.ReLU(), nn.Linear(2, 2), nn.Linear(2, 3))
    assert len(flat_parameters(seq)) == 6
    del seq[2]
    assert [id(p) for p in flat_parameters(seq)] == [id(p) for p in seq.parameters()] and len(flat_parameters(seq)) == 4
    seq.pop(2)
    assert [id(p) for p in flat_parameters(seq)] == [id(p) for p in seq.parameters()] and len(flat_parameters(seq)) == 2
    outer = nn.Sequential(nn.Sequential(nn.Linear(2, 2), nn.Linear(2, 2)), nn.Linear(2, 2))
    assert len(flat_parameters(outer)) == 6
    delattr(outer[0], '1')                                             # a grandchild
    assert [id(p) for p in flat_parameters(outer)] == [id(p) for p in outer.parameters()] and len(flat_parameters(outer)) == 4


def test_argument_structs_match_the_header_byte_for_byte(tmp_path):
    """lib.MergeArgs / lib.CloudArgs are ctypes mirrors of DclrMergeArgs / DclrCloudArgs (include/deepclr_amd.h): compile
    the header with gcc and compare sizes and the offsets of the last members (a field added on one side only would shift
    every pointer behind it)."""
    import subprocess
    from deepclr_amd import lib
    src = tmp_path / 'sizes.c'
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "deepclr_amd.h"\n'
                   'int main(void) { printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(DclrMergeArgs), offsetof(DclrMergeArgs, y), '
                   'offsetof(DclrMergeArgs, overflow), sizeof(DclrCloudArgs), offsetof(DclrCloudArgs, f_rows), '
                   'offsetof(DclrCloudArgs, merge)); return 0; }\n')
    exe = tmp_path / 'sizes'
    subprocess.run(['gcc', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)], check=True)
    got = [int(v) for v in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    m, c = lib.MergeArgs, lib.CloudArgs
    assert got == [ctypes.sizeof(m), m.y.offset, m.overflow.offset, ctypes.sizeof(c), c.f_rows.offset, c.merge.offset]


def test_batch_norm_and_dropout_configurations_build_load_and_fold():
    """`batch_norm: true`, `dropout` < 1 (reference helper.py:27-36,57-63,107-113; deepclr.py:63-70,260-261): the same module
    tree, hence a strict load of a reference-layout state_dict; in eval mode the running statistics fold into the weights the
    kernels pack (checked here against torch's own eval-mode modules on the CPU) and dropout is the identity; in training mode
    the inference entry points refuse."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import helpers
    from deepclr_amd.models import build_model
    from deepclr_amd.models.helper import flat_parameters
    cfg = helpers.small_bn_cfg()
    sd = synthetic.random_state_dict(cfg, seed=19)
    model = build_model(model_config_from_dict(cfg))
    assert set(model.state_dict()) == set(sd)
    model.load_state_dict(sd, strict=True)
    assert not set(model.state_dict()) <= set(synthetic.random_state_dict(helpers.small_cfg(), seed=19))   # really more keys
    model.eval()
    head = model._merge_layers[1]
    assert head.linear.has_dropout() and len(head.linear.layers()) == 2
    x = torch.randn(3, head.conv.layers()[0].affine.in_channels, 17)
    h = x
    for w, b in head.conv.affine_params():
        h = torch.relu(torch.nn.functional.conv1d(h, w, b))
    torch.testing.assert_close(h, head.conv.forward_torch(x), rtol=1e-5, atol=1e-6)
    g = h.max(dim=2)[0]
    want = head.linear.forward_torch(g)                                   # eval: dropout = identity
    for w, b in head.linear.affine_params():
        g = torch.relu(torch.nn.functional.linear(g, w, b))
    torch.testing.assert_close(g, want, rtol=1e-5, atol=1e-6)
    unit = model._cloud_layers[0]._sa0.mlps[0].layer0
    assert unit.conv.bias is None
    y = torch.randn(2, unit.conv.in_channels, 5, 7)
    w, b = unit.folded()
    torch.testing.assert_close(torch.relu(torch.nn.functional.conv2d(y, w, b)), unit(y.clone()), rtol=1e-5, atol=1e-6)
    # the running statistics are part of what the packed weights depend on: a change must invalidate them
    key = lambda: tuple((p.data_ptr(), p._version) for p in flat_parameters(head.conv))      # noqa: E731
    before = key()
    head.conv.layers()[0].norm.running_mean.add_(0.1)
    assert key() != before
    # training mode: batch statistics / random masks -- not what the inference kernels compute
    model.train()
    with pytest.raises(RuntimeError, match='RUNNING statistics'):
        head.conv.affine_params()
    with torch.no_grad(), pytest.raises(RuntimeError, match='model.eval'):
        model(torch.zeros(2, 64, 4))


def test_replaced_submodule_is_seen_by_the_parameter_slots():
    """flat_parameters caches (owner dict, name) slots per tree shape; replacing a submodule by one of the same arity must
    rebuild them (the stamp counted entries only and the packed weights stayed those of the replaced layer)."""
    from deepclr_amd.models.helper import Conv1d, Conv1dMultiLayer, flat_parameters
    m = Conv1dMultiLayer([4, 8, 6])
    old = m._sequential[1]
    assert any(p is old.affine.weight for p in flat_parameters(m))
    m._sequential[1] = Conv1d(8, 6)
    now = flat_parameters(m)
    assert not any(p is old.affine.weight for p in now) and any(p is m._sequential[1].affine.weight for p in now)
