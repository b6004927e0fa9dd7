"""The C restatement (oracle/als_oracle.c) against the numpy oracle and the
vectors produced by the upstream reference."""
import numpy as np
import pytest

from oracle import als_oracle as ao
from oracle import c_oracle as co
from oracle import imc_oracle as io
from problems import random_problem


@pytest.mark.parametrize('rank', [3, 10, 64])
def test_c_half_sweep_and_sse(rank):
    P = random_problem(40 + rank, 90, 70, 1500, rank, empty_rows=(0, 5), long_row=(7, 60))
    want = ao.half_sweep(P['indptr'], P['indices'], P['vals'], P['V'], rank, 0.1)
    got = co.half_sweep(P['indptr'], P['indices'], P['vals'], P['V'].T, 0.1, threads=2)
    np.testing.assert_allclose(got.T, want, rtol=1e-9, atol=1e-12)
    part = co.half_sweep(P['indptr'], P['indices'], P['vals'], P['V'].T, 0.1, 10, 20)
    np.testing.assert_array_equal(part, got[10:20])
    s = co.sse(P['indptr'], P['indices'], P['vals'], P['U'].T, P['V'].T)
    np.testing.assert_allclose(s, ao.train_sse(P['indptr'], P['indices'], P['vals'], P['U'],
                                               P['V']), rtol=1e-12)


def test_c_half_sweep_against_reference_vectors(golden):
    g = golden('als_c1')
    n, m = (int(v) for v in g['shape'])
    csr = ao.build_csr(g['users'], g['items'], g['vals'], (n, m))
    u1 = co.half_sweep(*csr, g['init_item_feats'].T, 0.1)
    np.testing.assert_allclose(u1.T, g['user_feats_1'], rtol=1e-9, atol=1e-12)
    csc = ao.csr_to_csc(n, m, *csr)
    i1 = co.half_sweep(*csc, u1, 0.1)
    np.testing.assert_allclose(i1.T, g['item_feats_1'], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(np.sqrt(co.sse(*csr, u1, i1) / csr[0][-1]), g['rmse'][0],
                               rtol=1e-11)


def test_c_singular_row_is_zero():
    indptr = np.array([0, 2, 2], dtype=np.int64)
    out = co.half_sweep(indptr, np.array([0, 1], dtype=np.int32), np.array([1.0, 2.0]),
                        np.zeros((2, 4)), 0.0)
    assert not out.any()


def test_c_imc_eval(golden):
    g = golden('imc_random')
    X, Y, users, items, vals = g['X'], g['Y'], g['users'], g['items'], g['vals']
    Z, S, lam = g['Z'], g['S'], float(g['lam'])
    indptr = np.zeros(X.shape[0] + 1, dtype=np.int64)
    np.cumsum(np.bincount(users, minlength=X.shape[0]), out=indptr[1:])
    G, sse = co.imc_eval(0, indptr, items, vals, X, Y, Z, threads=2)
    np.testing.assert_allclose((G + lam * Z).ravel(), g['grad'], rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(0.5 * (sse + vals.size * lam * np.sum(Z * Z)), g['cost'],
                               rtol=1e-12)
    H, _ = co.imc_eval(1, indptr, items, vals, X, Y, S)
    np.testing.assert_allclose((H + lam).ravel(), g['hessp'], rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(io.grad_indexed(Z, X, Y, users, items, vals, lam), G + lam * Z,
                               rtol=1e-10, atol=1e-10)
