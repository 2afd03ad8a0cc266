"""Write synthetic clouds (deepclr_amd/synthetic.py) as raw float32 (b, n, 4) for the C++ probe harnesses."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from deepclr_amd import synthetic
kind, b, n, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
x = synthetic.make_batch(kind, (b + 1) // 2, n)[:b]
assert x.shape == (b, n, 4)
x.astype(np.float32).tofile(out)
print('wrote', out, x.shape)
