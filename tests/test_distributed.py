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
