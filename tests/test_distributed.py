"""World-size-2 gloo tests of the pair sharding + output gather (CPU; the compute function is the oracle)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from deepclr_amd import distributed as D
from deepclr_amd import synthetic
from helpers import small_cfg


def test_pair_range_partitions_everything():
    for n, world in ((8, 2), (64, 8), (7, 2), (5, 8), (1, 3)):
        spans = [D.pair_range(n, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1


def test_local_batch_layout():
    x = torch.arange(2 * 6).float().view(12, 1, 1)
    assert D.local_batch(x, 1, 3)[:, 0, 0].tolist() == [2, 3, 8, 9]


def _worker(rank, world, port, n_pairs, tmp):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import oracle
    torch.set_num_threads(2)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    cfg = small_cfg()
    orc = oracle.build_oracle_model(cfg, synthetic.random_state_dict(cfg, 3))
    x = torch.from_numpy(synthetic.make_batch('kitti', n_pairs, 256))
    y = D.sharded_forward(orc, x)
    torch.save(y, os.path.join(tmp, 'y%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('n_pairs', [4, 3])
def test_sharded_forward_matches_single_process(tmp_path, n_pairs):
    import oracle
    port = 29500 + (os.getpid() % 2000) + n_pairs
    mp.spawn(_worker, args=(2, port, n_pairs, str(tmp_path)), nprocs=2, join=True)
    cfg = small_cfg()
    orc = oracle.build_oracle_model(cfg, synthetic.random_state_dict(cfg, 3))
    want = orc(torch.from_numpy(synthetic.make_batch('kitti', n_pairs, 256)))
    for r in range(2):
        got = torch.load(os.path.join(str(tmp_path), 'y%d.pt' % r), weights_only=True)
        assert got.shape == want.shape
        np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-6, atol=1e-7)


_RCCL_WORKER = r"""
import os, sys, json
sys.path.insert(0, sys.argv[1])
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=sys.argv[2], RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.join(sys.argv[1], 'tests'))
from helpers import small_cfg
from deepclr_amd import distributed as D, synthetic
from deepclr_amd.config import model_config_from_dict
from deepclr_amd.models import build_model
import bench
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
dist.init_process_group('nccl', device_id=dev)                      # nccl == RCCL on ROCm
cfg = small_cfg()
model = build_model(model_config_from_dict(cfg))
model.load_state_dict(synthetic.random_state_dict(cfg, 3))
model = model.to(dev).eval()
x = torch.from_numpy(synthetic.make_batch('kitti', 4, 512)).to(dev)
with torch.no_grad():
    model(x.clone())                                                  # (the checkpoint's first forward is the range-checked one)
    y_plain, _, _ = model(x.clone())
    y_shard = D.sharded_forward(lambda b: model(b)[0], x.clone())    # local_batch + forward + all_gather_into_tensor
    y_gath = D.gather_outputs(y_plain, 4)
gather = bench.OutputGather(dist, 1, 2, 4, y_plain.shape[1], dev)   # bench.py's N > 1 bookkeeping: two steps per collective
gather.put(y_plain)
gather.slot().copy_(2.0 * y_plain)
gather.put(None)
torch.cuda.synchronize()
backend = dist.get_backend()
print(json.dumps({'backend': backend, 'world': dist.get_world_size(), 'collectives': gather.collectives,
                  'shard_equal': bool(torch.equal(y_shard, y_plain)), 'gather_equal': bool(torch.equal(y_gath, y_plain)),
                  'bench_equal': bool(torch.equal(gather.gathered[:4], y_plain) and torch.equal(gather.gathered[4:], 2.0 * y_plain)),
                  'finite': bool(torch.isfinite(y_plain).all())}))
dist.barrier()
dist.destroy_process_group()
"""


@pytest.mark.gpu
def test_rccl_one_rank_group_gathers_hip_outputs(tmp_path):
    """RCCL in front of the driver. A FRESH child process initialises a one-rank `nccl` (= RCCL) group on
    the GPU, runs the sharded forward, gather_outputs and bench.py's OutputGather on HIP outputs of a small model, and the
    gathered poses equal the plain forward's bit for bit (one all-gather of (B_local, 8) is the only exchange)."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / 'rccl_worker.py'
    script.write_text(_RCCL_WORKER)
    port = str(29500 + (os.getpid() % 2000) + 17)
    res = subprocess.run([sys.executable, str(script), root, port], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    doc = json.loads([l for l in res.stdout.splitlines() if l.startswith('{')][-1])
    assert doc == {'backend': 'nccl', 'world': 1, 'collectives': 1, 'shard_equal': True, 'gather_equal': True,
                   'bench_equal': True, 'finite': True}, doc
