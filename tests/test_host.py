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
    for mutate in (lambda c: c['params'].update(batch_norm=True),
                   lambda c: c['params']['merge']['params'].update(k=0),
                   lambda c: c['params']['merge']['params'].update(k=40),
                   lambda c: c['params']['merge']['params'].update(mlp=[64, 64, 128]),
                   lambda c: c['params']['cloud_features']['params'].update(mlps=[[[32, 32, 64], [16, 16, 32]]])):
        cfg = synthetic.model_cfg('kitti')
        mutate(cfg)
        with pytest.raises(NotImplementedError):
            build_model(model_config_from_dict(cfg))
    assert base == synthetic.model_cfg('kitti')


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
    with pytest.raises(NotImplementedError):
        LabelType.POSE3D_EULER.to_matrix(np.zeros(6))


def test_synthetic_inputs_are_deterministic():
    a = synthetic.make_batch('kitti', 2, 128)
    b = synthetic.make_batch('kitti', 2, 128)
    assert a.shape == (4, 128, 4) and a.dtype == np.float32 and np.array_equal(a, b)
    assert not np.array_equal(a[0], synthetic.make_batch('kitti', 1, 128, first_pair=1)[0])
    m = synthetic.make_batch('modelnet', 1, 64)
    assert m.shape == (2, 64, 3) and np.abs(m).max() < 1.5


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
    with pytest.raises(RuntimeError):
        load_config('cfg.yaml', Mode.TEST)
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
