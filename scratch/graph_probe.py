"""Single-pair forward as one hipGraph: does capture work through the ctypes launches, and what does it save?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepclr_amd import synthetic
from deepclr_amd.config import model_config_from_dict
from deepclr_amd.models import build_model
dev = torch.device('cuda', 0)
cfg = synthetic.model_cfg('kitti')
model = build_model(model_config_from_dict(cfg)); model.load_state_dict(synthetic.random_state_dict(cfg, seed=0)); model = model.to(dev).eval()
for pairs in (1, 8):
    x = torch.from_numpy(synthetic.make_batch('kitti', pairs, 16384)).to(dev)
    with torch.no_grad():
        for _ in range(3): y_ref = model(x)[0]
    torch.cuda.synchronize()
    def timed(fn, n=50):
        torch.cuda.synchronize(); ts = []
        for _ in range(n):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
        ts.sort(); return ts[len(ts) // 2]
    with torch.no_grad():
        eager = timed(lambda: model(x))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    xs = x.clone()
    try:
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(2): model(xs)
        torch.cuda.current_stream().wait_stream(side)
        with torch.cuda.graph(g), torch.no_grad():
            ys = model(xs)[0]
        torch.cuda.synchronize()
        g.replay(); torch.cuda.synchronize()
        same = torch.equal(ys, y_ref)
        graphed = timed(g.replay)
        print('%d pair(s): eager %.1f us, graph replay %.1f us, identical %s' % (pairs, eager * 1e3, graphed * 1e3, same))
    except Exception as ex:
        print('%d pair(s): eager %.1f us, capture failed: %r' % (pairs, eager * 1e3, ex))
