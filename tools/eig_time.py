import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import numpy as np, vio_ct
from test_gpu_eig import _eig
P = vio_ct.pkg()
for n in (76, 104, 136, 160, 185, 197, 320, 495):
    rng = np.random.default_rng(n); B = rng.standard_normal((n, n)); A = B @ B.T
    w, V, us = _eig(P, A)
    w, V, us = _eig(P, A)
    print(n, [round(float(x) / 1e3, 2) for x in us], 'ms (total, tridiagonalisation, accumulation, QL)')
