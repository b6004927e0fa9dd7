import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'recommendation-models_amd'))
import numpy as np
from alsmi import IMC
rng = np.random.default_rng(0)
n, m, p, q = 1000, 26744, 128, 256
X = rng.standard_normal((n, p)); Y = rng.standard_normal((m, q))
est = IMC(alpha=0.1); est.Z = rng.standard_normal((p, q)); est.n_components_ = p
est.predict_all((X[:10], Y))
t0 = time.perf_counter(); P = est.predict_all((X, Y)); dt = time.perf_counter() - t0
want = X[:50].dot(est.Z).dot(Y.T)
print('predict_all %d x %d: %.3f s, max rel diff %.2e' % (n, m, dt, np.abs(P[:50] - want).max() / np.abs(want).max()))
print('predict_one', est.predict_one((X[3], Y[7])), X[3].dot(est.Z).dot(Y[7]))
