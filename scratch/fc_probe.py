"""The fully connected tail kernel alone: time per launch, and (round 5) bit-identity of a changed kernel against another
build of the library (DCLR_LIB_OTHER)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from deepclr_amd import lib
dev = 'cuda:0'
other = ctypes.CDLL(os.environ['DCLR_LIB_OTHER']) if os.environ.get('DCLR_LIB_OTHER') else None
g = torch.Generator().manual_seed(0)
for m in (1, 3, 8, 9, 80, 256):
    for k, n, act in ((1024, 512, 1), (512, 256, 1), (256, 8, 2), (250, 7, 3), (1024, 512, 0)):
        x = torch.randn(m, k, generator=g).to(dev); w = (torch.randn(n, k, generator=g) * 0.05).to(dev); b = torch.randn(n, generator=g).to(dev)
        y = torch.empty(m, n, device=dev); y2 = torch.empty(m, n, device=dev)
        call = lambda L, out: L.dclr_fc(m, n, k, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(b.data_ptr()), act, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(0))
        assert call(lib.load(), y) == 0
        torch.cuda.synchronize()
        same = ''
        if other is not None:
            assert call(other, y2) == 0
            torch.cuda.synchronize()
            same = 'bit-identical to the other build' if torch.equal(y, y2) else 'DIFFERS from the other build by %.3g' % float((y - y2).abs().max())
        ts = []
        for _ in range(10):
            s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); call(lib.load(), y); t.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(t) * 1e3)
        print('m %3d k %4d n %3d act %d: median %6.1f us  %s' % (m, k, n, act, float(np.median(ts)), same), flush=True)
