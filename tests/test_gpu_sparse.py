"""Device-side structure builders against scipy / the oracle: bit-exact index arrays."""
import numpy as np
import pytest
import scipy.sparse as sps

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def gpu():
    from alsmi import device as dev
    return dev.require_gpu()


def lil_reference(users, items, y, shape):
    ref = sps.lil_matrix(shape)
    for idx, (i, j) in enumerate(zip(users, items)):
        ref[i, j] = y[idx]
    return ref.tocsr()


@pytest.mark.parametrize('dtype', ['float64', 'float32'])
def test_build_csr_lil_semantics(gpu, dtype):
    """duplicates (last write wins), zeros (dropped, also when they overwrite), negative
    indices, empty rows — compared with the reference's literal lil fill."""
    from alsmi import device as dev
    rng = np.random.default_rng(3)
    n, m, n_in = 23, 17, 400
    users = rng.integers(0, n, n_in)
    items = rng.integers(0, m, n_in)
    users[users == 5] = 6                      # an empty row
    y = rng.choice([0.0, 0.5, 1.0, 2.5, 4.0], size=n_in)
    users[:3], items[:3], y[:3] = [1, 1, 1], [2, 2, 2], [3.0, 0.0, 4.5]     # last wins
    users[3:5], items[3:5], y[3:5] = [2, 2], [7, 7], [1.5, 0.0]             # zero overwrites
    users[5], items[5] = -1, -2                                            # numpy-style negatives
    ref = lil_reference(np.where(users < 0, users + n, users), np.where(items < 0, items + m, items),
                        y, (n, m))
    indptr, indices, vals = dev.build_csr(users, items, y, (n, m), dtype, gpu)
    assert np.array_equal(indptr.cpu().numpy(), ref.indptr)
    assert np.array_equal(indices.cpu().numpy(), ref.indices)
    assert np.array_equal(vals.cpu().numpy(), ref.data.astype(dtype))
    assert indptr.dtype.is_floating_point is False and int(indptr[6]) == int(indptr[5])


def test_build_csr_matches_oracle_medium(gpu):
    from alsmi import device as dev
    from oracle import als_oracle as ao
    rng = np.random.default_rng(4)
    n, m, n_in = 3000, 1700, 200000
    users, items = rng.integers(0, n, n_in), rng.integers(0, m, n_in)
    y = rng.integers(0, 6, n_in).astype(np.float64)        # many zeros and duplicates
    want = ao.build_csr(users, items, y, (n, m))
    got = dev.build_csr(users, items, y, (n, m), 'float64', gpu)
    for g, w in zip(got, want):
        assert np.array_equal(g.cpu().numpy(), w)
    # empty input
    e = np.zeros(0, dtype=np.int64)
    indptr, indices, vals = dev.build_csr(e, e, np.zeros(0), (5, 4), 'float64', gpu)
    assert not indptr.cpu().numpy().any() and indices.numel() == 0 and vals.numel() == 0


@pytest.mark.parametrize('dtype', ['float64', 'float32'])
def test_csr_to_csc_bit_exact(gpu, dtype):
    import torch
    from alsmi import device as dev
    rng = np.random.default_rng(5)
    n, m = 900, 1300
    mat = sps.random(n, m, density=0.02, format='csr', random_state=7, dtype=np.float64)
    mat.data = rng.integers(1, 11, mat.nnz) / 2.0
    mat[17] = 0                                  # an empty row
    mat.eliminate_zeros()
    mat.sort_indices()
    csc = mat.tocsc()
    to = lambda a, t: torch.from_numpy(np.ascontiguousarray(a)).to(gpu, t)
    td = torch.float32 if dtype == 'float32' else torch.float64
    cptr, cidx, cvals = dev.csr_to_csc(to(mat.indptr, torch.int64), to(mat.indices, torch.int32),
                                       to(mat.data, td), (n, m))
    assert np.array_equal(cptr.cpu().numpy(), csc.indptr)
    assert np.array_equal(cidx.cpu().numpy(), csc.indices)
    assert np.array_equal(cvals.cpu().numpy(), csc.data.astype(dtype))


def test_round_trip_and_full_size(gpu):
    """transpose twice = identity; the 20 M-entry benchmark matrix against torch's own sort."""
    import torch
    from alsmi import device as dev
    from alsmi import synth
    prob = synth.sharded_problem(20242, 138493, 26744, 20000000, 1, 0, gpu)
    indptr, indices, vals = prob['csr']
    n, m = prob['shape']
    cptr, cidx, cvals = dev.csr_to_csc(indptr, indices, vals, (n, m))
    want = prob['csc']
    assert torch.equal(cptr, want[0]) and torch.equal(cidx, want[1]) and torch.equal(cvals, want[2])
    back = dev.csr_to_csc(cptr, cidx, cvals, (m, n))
    assert torch.equal(back[0], indptr) and torch.equal(back[1], indices) and torch.equal(back[2], vals)
