"""Result-file format and odometry metrics against fixtures written by the reference's evaluation code
(tests/golden/make_eval_golden.py)."""
import os

import numpy as np

from deepclr_amd import evaluation as ev

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _expected():
    return np.load(os.path.join(GOLDEN, 'eval_expected.npz'))


def test_reads_reference_file_and_rewrites_it_identically(tmp_path):
    src = os.path.join(GOLDEN, 'eval_sequence.txt')
    seq = ev.Sequence.read(src)
    assert len(seq) == 700 and seq.table().shape == (700, 26)
    out = tmp_path / 'kitti_00.txt'
    seq.write(str(out))
    assert out.read_text() == open(src).read()                      # same numbers, same text formatting
    again = ev.Evaluator.read(str(tmp_path))
    assert again.has_sequence('kitti_00') and len(again.get_sequence('kitti_00')) == 700


def test_pose_chain_and_path_length_match_reference():
    want = _expected()
    seq = ev.Sequence.read(os.path.join(GOLDEN, 'eval_sequence.txt'))
    np.testing.assert_allclose(ev.chain_poses(seq.prediction), want['poses_pred'], rtol=0, atol=1e-9)
    np.testing.assert_allclose(ev.chain_poses(seq.ground_truth), want['poses_gt'], rtol=0, atol=1e-9)
    np.testing.assert_allclose(ev.travelled(seq.ground_truth), want['distances_gt'], rtol=0, atol=1e-9)


def test_step_and_segment_errors_match_reference():
    want = _expected()
    seq = ev.Sequence.read(os.path.join(GOLDEN, 'eval_sequence.txt'))
    step = ev.step_errors(seq)
    np.testing.assert_allclose(step['translation'], want['step_translation'], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(step['rotation'], want['step_rotation'], rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(step['translation_rmse'], want['step_translation_rmse'], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(step['rotation_chordal'], want['step_rotation_chordal'], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(step['time'], want['step_time'])
    seg = ev.segment_errors(seq)
    assert np.array_equal(seg['first_frame'], want['seg_first'])
    assert np.array_equal(seg['segment_length'], want['seg_length'])
    np.testing.assert_allclose(seg['speed'], want['seg_speed'], rtol=1e-12)
    np.testing.assert_allclose(seg['translation'], want['seg_translation'], rtol=1e-8, atol=1e-14)
    np.testing.assert_allclose(seg['rotation'], want['seg_rotation'], rtol=1e-6, atol=1e-12)


def test_evaluator_collects_sequences_and_skips_the_stateless_first_frame(tmp_path):
    e = ev.Evaluator()
    e.add_transforms('a', 0.0, None, np.eye(4), 1.0)                 # sequential mode: no prediction yet
    t = np.eye(4)
    t[0, 3] = 1.0
    for i in range(5):
        e.add_transforms('a', 0.1 * i, t, t, 2.0)
    e.add_transforms('b', 0.0, np.eye(4), t, 3.0)
    assert list(e.get_sequences()) == ['a', 'b'] and len(e.get_sequence('a')) == 5
    e.write(str(tmp_path))
    assert sorted(os.listdir(str(tmp_path))) == ['a.txt', 'b.txt']
    s = ev.Evaluator.read(str(tmp_path)).summary()
    assert abs(s['step_translation_mean [m]'] - 1.0 / 6) < 1e-12      # five exact pairs, one off by 1 m
    assert abs(s['time_mean [ms]'] - 13.0 / 6) < 1e-12
    assert np.isnan(s['kitti_translation [%]'])                      # 5 m driven: no 100 m segment exists
