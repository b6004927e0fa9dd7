"""Synthetic rating matrices with MovieLens-like power-law degrees (torch, any device).

Used by bench.py and the large-shape tests: there is no network for real
datasets.  A problem is generated in user BLOCKS so that every rank of a
multi-GPU run can rebuild exactly the blocks it needs from (seed, block id).
"""
import numpy as np
import torch


def _item_cdf(n_items, shift, device):
    w = 1.0 / (torch.arange(n_items, dtype=torch.float64, device=device) + shift)
    cdf = torch.cumsum(w, 0)
    return cdf / cdf[-1]


def user_block(seed, block, n_users, n_items, nnz, device, planted_rank=16, d_min=20,
               d_max=10000, zipf_a=1.2, item_shift=60.0):
    """One block of `n_users` users with exactly `nnz` distinct (user, item) entries.

    User degrees follow a power law (Pareto tail exponent `zipf_a`, clipped to
    [d_min, d_max]); items are drawn with popularity ~ 1/(rank + item_shift), the
    popularity ranks scattered over item ids by a seed-only permutation (shared by
    all blocks); ratings are clip(round(2*(3.5 + <p_u, q_i> + N(0, .5)))/2, .5, 5)
    with planted factors of rank `planted_rank`, so they are never zero.

    Returns (users_local int64 [nnz], items int64 [nnz], vals float32 [nnz]) sorted
    by (user, item)."""
    g = torch.Generator(device=device)
    g.manual_seed(int(seed) * 1000003 + 7919 * int(block) + 1)
    gi = torch.Generator(device=device)
    gi.manual_seed(int(seed) * 1000003)  # item-side randomness is block independent
    perm = torch.randperm(n_items, generator=gi, device=device)
    q = torch.randn((n_items, planted_rank), generator=gi, device=device) * 0.418
    cdf = _item_cdf(n_items, item_shift, device)
    d_max = min(d_max, n_items)
    u = torch.rand(n_users, generator=g, device=device, dtype=torch.float64)
    base = d_min * (1.0 - u).clamp_min(1e-12) ** (-1.0 / zipf_a)
    over = 1.12
    for _ in range(8):
        # scale the degrees so that the oversampled total lands above nnz after dedup
        lo_s, hi_s = 1e-3, 1e3
        for _ in range(60):
            mid = (lo_s * hi_s) ** 0.5
            tot = float(torch.clamp(torch.round(base * mid), d_min, d_max).sum())
            if tot < nnz * over:
                lo_s = mid
            else:
                hi_s = mid
        deg = torch.clamp(torch.round(base * hi_s), d_min, d_max).to(torch.int64)
        users = torch.repeat_interleave(torch.arange(n_users, device=device), deg)
        r = torch.rand(users.numel(), generator=g, device=device, dtype=torch.float64)
        ranks = torch.searchsorted(cdf, r).clamp_max(n_items - 1)
        key = torch.unique(users * n_items + perm[ranks])
        if key.numel() >= nnz:
            break
        over *= 1.15
    else:
        raise RuntimeError('could not reach the requested number of distinct entries')
    if key.numel() > nnz:
        keep = torch.randperm(key.numel(), generator=g, device=device)[:nnz]
        key = key[torch.sort(keep).values]
    users = key // n_items
    items = key % n_items
    p = torch.randn((n_users, planted_rank), generator=g, device=device) * 0.418
    noise = torch.randn(nnz, generator=g, device=device) * 0.5
    dots = torch.empty(nnz, device=device)
    step = 1 << 22
    for s in range(0, nnz, step):
        e = min(nnz, s + step)
        dots[s:e] = (p[users[s:e]] * q[items[s:e]]).sum(-1)
    vals = torch.clamp(torch.round(2.0 * (3.5 + dots + noise)) / 2.0, 0.5, 5.0)
    return users, items, vals.to(torch.float32)


def csr_from_sorted(rows, n_rows):
    counts = torch.bincount(rows, minlength=n_rows)
    indptr = torch.zeros(n_rows + 1, dtype=torch.int64, device=rows.device)
    torch.cumsum(counts, 0, out=indptr[1:])
    return indptr


def item_bounds_by_popularity(seed, n_items, parts, device, item_shift=60.0):
    """Contiguous item-id ranges with (almost) equal EXPECTED numbers of entries: an item's
    share of the entries is its popularity weight 1/(rank + item_shift), the ranks scattered
    over the ids by the seed's permutation (see user_block), so every rank derives the same
    bounds without seeing the other ranks' blocks.  int64 array of parts+1 item ids."""
    gi = torch.Generator(device=device)
    gi.manual_seed(int(seed) * 1000003)
    perm = torch.randperm(n_items, generator=gi, device=device)
    w_rank = 1.0 / (torch.arange(n_items, dtype=torch.float64, device=device) + item_shift)
    w_id = torch.empty_like(w_rank)
    w_id[perm] = w_rank                      # weight of item id perm[rank]
    cum = torch.cumsum(w_id, 0)
    targets = cum[-1] * torch.arange(1, parts, dtype=torch.float64, device=device) / parts
    cuts = torch.searchsorted(cum, targets).cpu().numpy().astype(np.int64)
    bounds = np.concatenate(([0], cuts, [n_items])).astype(np.int64)
    return np.maximum.accumulate(np.minimum(bounds, n_items))


def sharded_problem(seed, n_users_per_block, n_items, nnz_per_block, world, me, device,
                    n_blocks=None):
    """A problem of `n_blocks` user blocks (default `world`: weak scaling, one block per
    rank; a fixed n_blocks that `world` divides: strong scaling, n_blocks/world consecutive
    blocks per rank): this rank's CSR slice (its own blocks) and CSC slice (its item range,
    cut by expected entries, over every block).

    Returns dict(shape, csr, csc, user_bounds, item_bounds, total_nnz); csr/csc are
    (indptr int64, indices int32, vals float32) device tensors with LOCAL row ids and
    GLOBAL column ids."""
    n0, m = int(n_users_per_block), int(n_items)
    n_blocks = int(world if n_blocks is None else n_blocks)
    if n_blocks % world:
        raise ValueError('the number of ranks (%d) must divide the number of blocks (%d)'
                         % (world, n_blocks))
    per = n_blocks // world
    item_bounds = item_bounds_by_popularity(seed, m, world, device) if world > 1 else \
        np.array([0, m], dtype=np.int64)
    ilo, ihi = int(item_bounds[me]), int(item_bounds[me + 1])
    r_users, r_items, r_vals = [], [], []
    c_items, c_users, c_vals = [], [], []
    for b in range(n_blocks):
        users, items, vals = user_block(seed, b, n0, m, nnz_per_block, device)
        if me * per <= b < (me + 1) * per:
            r_users.append(users + (b - me * per) * n0)
            r_items.append(items.to(torch.int32))
            r_vals.append(vals)
        if world == 1:
            sel = slice(None)
            c_items.append(items.to(torch.int32))
        else:
            sel = (items >= ilo) & (items < ihi)
            c_items.append((items[sel] - ilo).to(torch.int32))
        c_users.append((users[sel] + b * n0).to(torch.int32))
        c_vals.append(vals[sel])
        del users, items, vals, sel
    csr = (csr_from_sorted(torch.cat(r_users), n0 * per), torch.cat(r_items), torch.cat(r_vals))
    del r_users, r_items, r_vals
    ci, cu, cv = torch.cat(c_items), torch.cat(c_users), torch.cat(c_vals)
    del c_items, c_users, c_vals
    # stable sort by item keeps the users of an item ascending (the blocks were appended in
    # user order)
    ci, order = torch.sort(ci.to(torch.int64), stable=True)
    cu, cv = cu[order], cv[order]
    del order
    csc = (csr_from_sorted(ci, ihi - ilo), cu, cv)
    return dict(shape=(n0 * n_blocks, m), csr=csr, csc=csc,
                user_bounds=np.arange(world + 1, dtype=np.int64) * (n0 * per),
                item_bounds=item_bounds, total_nnz=int(nnz_per_block) * n_blocks)


def init_factors(seed, rank, n_items, item_sums, item_counts):
    """The reference's initial item factors (src/als.py:147-150) for a synthetic
    problem: uniform [0,1) entries from numpy's RandomState, row 0 = mean rating."""
    rs = np.random.RandomState(seed)
    feats = rs.rand(rank, n_items)
    with np.errstate(invalid='ignore', divide='ignore'):
        avg = item_sums / item_counts
    avg[np.isnan(avg)] = 0
    feats[0] = avg
    return feats
