#!/usr/bin/env python3
"""Lane-level numpy model of the in-register block solver of als_sweep.hip (fp32 path).

Design aid, not product code: it mirrors the HIP routine instruction group by instruction
group (64 lanes, four accumulator registers per 16x16 tile, ds_bpermute, the 4x4x1 and
16x16x4 matrix instructions, DPP row reductions) so that index maps and signs can be checked
on the CPU against numpy.linalg.solve before the kernel is run.
"""
import sys

import numpy as np

LANES = np.arange(64)
G, Q = LANES >> 4, LANES & 15


def crow(g, r):
    return 4 * g + r


def mfma16(X, Y, C):
    """C + X^T Y, all three in accumulator layout: lane (g,q), reg r = M[4g+r][q]."""
    def dense(T):
        M = np.zeros((16, 16), T.dtype)
        for lane in range(64):
            for r in range(4):
                M[crow(lane >> 4, r), lane & 15] = T[lane, r]
        return M
    D = dense(X).T.dot(dense(Y)).astype(C.dtype)
    out = C.copy()
    for lane in range(64):
        for r in range(4):
            out[lane, r] += D[crow(lane >> 4, r), lane & 15]
    return out


def mfma4(a, b, C):
    """16 blocks of 4x4x1: block = lane // 4; D[lane][r] += a[4*block + r] * b[lane]."""
    out = C.copy()
    for lane in range(64):
        blk = lane // 4
        for r in range(4):
            out[lane, r] += a[4 * blk + r] * b[lane]
    return out


def bperm(src, src_lane):
    return src[src_lane]


def row_allreduce(v):
    """sum over the 16 lanes of each row, result in every lane of the row."""
    out = v.copy()
    for g in range(4):
        out[16 * g:16 * g + 16] = v[16 * g:16 * g + 16].sum(0)
    return out


def col_allreduce(v):
    """sum over the four rows (same q), result in every lane."""
    s = v[0:16] + v[16:32] + v[32:48] + v[48:64]
    return np.concatenate([s, s, s, s])


def sweep_invert(a, dtype):
    """In-place symmetric sweep of all 16 pivots: tile -> -inverse.  Returns (tile, ok)."""
    ok = True
    for p in range(16):
        gp, rp = p // 4, p % 4
        src = a[:, rp]
        rowq = bperm(src, 16 * gp + Q)                    # A[p][q]
        colA = bperm(src, 16 * gp + 4 * G + (Q & 3))      # A[p][4g + (q&3)] = A[4g+(q&3)][p]
        d = src[16 * gp + p]
        ok = ok and bool(d > 0)
        inv = dtype(1) / d
        is_pc = Q == p
        a_op = colA * inv
        b_op = np.where(is_pc, dtype(0), -rowq)
        s = np.where(is_pc, inv, dtype(1))
        a = a * s[:, None]
        a = mfma4(a_op, b_op, a)
        t = np.where(is_pc, dtype(-1), rowq) * inv
        a[:, rp] = np.where(G == gp, t, a[:, rp])
    return a, ok


def block_solve(tiles, ycol, NV, dtype=np.float32):
    """tiles[(e,f)] (e <= f): [64,4] accumulator layout of block (e,f) of the permuted matrix
    (ridge included); ycol[e]: [64] column layout, lane (.,q) = b_e[q].  Returns xr[k]: [64,4]
    row layout, lane (g,.) reg r = x_k[4g+r]."""
    S = {k: v.astype(dtype).copy() for k, v in tiles.items()}
    y = [v.astype(dtype).copy() for v in ycol]
    tr = [None] * NV
    ok = True
    for k in range(NV):
        S[(k, k)], good = sweep_invert(S[(k, k)], dtype)
        ok = ok and good
        Nk = S[(k, k)]
        # tr = Nk y_k, column -> row layout: lane sums its row entries times y[q], row all-reduce
        tr[k] = row_allreduce(Nk * y[k][:, None])
        Tn = {}
        for J in range(k + 1, NV):
            Tn[J] = mfma16(Nk, S[(k, J)], np.zeros((64, 4), dtype))
        for I in range(k + 1, NV):
            for J in range(I, NV):
                S[(I, J)] = mfma16(S[(k, I)], Tn[J], S[(I, J)])
            # y_I += S_kI^T tr : row -> column layout
            part = (S[(k, I)] * tr[k]).sum(1)
            y[I] = y[I] + col_allreduce(part)
        for J in range(k + 1, NV):
            S[(k, J)] = Tn[J]
    xr = [None] * NV
    xc = [None] * NV
    for k in range(NV - 1, -1, -1):
        acc = np.zeros((64, 4), dtype)
        for J in range(k + 1, NV):
            acc = acc + S[(k, J)] * xc[J][:, None]
        xr[k] = -tr[k] + row_allreduce(acc)
        # row -> column layout: lane (.,q) takes x[q] from lane (q/4, .) reg q%4
        cand = [bperm(xr[k][:, r], 16 * (Q // 4) + Q) for r in range(4)]
        xc[k] = np.choose(Q % 4, cand)
    return xr, ok


def swizzle_lane15(v):
    """every lane reads lane 15 of its row (a ds_bpermute with address 4*(16g+15) in the kernel)."""
    return v[(LANES & 0x30) | 15]


def block_solve_ldl(tiles, ycol, NV, dtype=np.float32):
    """Stable variant: block LDL^T.  Per pivot block the sixteen elimination steps are applied,
    as row operations, to the whole block row [S_kk | S_kJ ... | R] (R starts as the identity
    and ends as inv(L_kk); its last column, which no row operation changes, carries the
    right-hand side), every tile updated by one 4x4x1 instruction per step; the trailing blocks
    then get S_IJ -= Z_I^T D^-1 Z_J on the 16x16x4 instruction.  No inverse of a Schur
    complement is ever formed.  Returns xc[k]: [64] column layout, lane (.,q) = x_k[q]."""
    S = {k: v.astype(dtype).copy() for k, v in tiles.items()}
    y = [v.astype(dtype).copy() for v in ycol]
    irow = 4 * G + (Q & 3)
    last_col = Q == 15
    rs, zr = [None] * NV, [None] * NV
    ok = True
    for k in range(NV):
        rt = np.zeros((64, 4), dtype)
        for r in range(4):
            yrow = bperm(y[k], 4 * G + r)
            rt[:, r] = np.where(last_col, yrow, np.where(4 * G + r == Q, 1, 0))
        row_tiles = [S[(k, k)]] + [S[(k, J)] for J in range(k + 1, NV)] + [rt]
        dv = np.ones((64, 4), dtype)
        for p in range(16):
            gp, rp = p // 4, p % 4
            a = row_tiles[0]
            cola = bperm(a[:, rp], 16 * gp + irow)  # A[p][i], i = 4g + q%4
            d = a[16 * gp + p, rp]
            ok = ok and bool(d > 0)
            inv = dtype(1) / d
            m = np.where(irow > p, -cola * inv, dtype(0))
            rows = [bperm(t[:, rp], 16 * gp + Q) for t in row_tiles]
            for t in range(len(row_tiles)):
                row_tiles[t] = mfma4(m, rows[t], row_tiles[t])
            dv[:, rp] = np.where(G == gp, inv, dv[:, rp])
        for j, J in enumerate(range(k + 1, NV)):
            S[(k, J)] = row_tiles[1 + j]  # Z_J = inv(L) S_kJ
        rt = row_tiles[-1]
        zr[k] = np.stack([swizzle_lane15(rt[:, r]) for r in range(4)], axis=1)
        dz = dv * zr[k]
        rtrue = rt.copy()
        for r in range(4):
            rtrue[:, r] = np.where(last_col, np.where(4 * G + r == 15, 1, 0), rt[:, r])
        rs[k] = dv * rtrue
        for I in range(k + 1, NV):
            zi = S[(k, I)]
            for J in range(I, NV):
                S[(I, J)] = mfma16(zi, -dv * S[(k, J)], S[(I, J)])
            y[I] = y[I] - col_allreduce((zi * dz).sum(1))
    xc = [None] * NV
    for k in range(NV - 1, -1, -1):
        w = np.zeros((64, 4), dtype)
        for J in range(k + 1, NV):
            w = w + S[(k, J)] * xc[J][:, None]
        rhs = zr[k] - row_allreduce(w)
        xc[k] = col_allreduce((rs[k] * rhs).sum(1))
    return xc, ok


def from_cols(xc, NV):
    x = np.zeros(16 * NV, xc[0].dtype)
    for k in range(NV):
        x[np.arange(16) * NV + k] = xc[k][:16]
    return x


def to_tiles(A, b, NV):
    """dense (KP x KP) SPD matrix and rhs in natural order -> accumulator-layout tiles of the
    permuted matrix: tile (e,f)[t][t'] = A[t*NV+e][t'*NV+f], and column-layout rhs."""
    tiles, ycol = {}, []
    for e in range(NV):
        for f in range(e, NV):
            T = np.zeros((64, 4), A.dtype)
            for lane in range(64):
                g, q = lane >> 4, lane & 15
                for r in range(4):
                    T[lane, r] = A[crow(g, r) * NV + e, q * NV + f]
            tiles[(e, f)] = T
        ycol.append(b[Q * NV + e].copy())
    return tiles, ycol


def from_rows(xr, NV):
    x = np.zeros(16 * NV, xr[0].dtype)
    for k in range(NV):
        for g in range(4):
            for r in range(4):
                x[crow(g, r) * NV + k] = xr[k][16 * g, r]
    return x


def main():
    rng = np.random.default_rng(5)
    worst = {}
    for NV in (1, 2, 4):
        KP = 16 * NV
        for trial in range(6):
            n = rng.integers(3, 300)
            F = rng.standard_normal((KP, n)) * 0.5 + 0.3
            A = F.dot(F.T) + 0.1 * n * np.eye(KP)
            b = F.dot(rng.integers(1, 6, n).astype(np.float64))
            want = np.linalg.solve(A, b)
            for dtype in (np.float64, np.float32):
                tiles, ycol = to_tiles(A.astype(dtype), b.astype(dtype), NV)
                xr, ok = block_solve(tiles, ycol, NV, dtype)
                got = from_rows(xr, NV).astype(np.float64)
                err = np.linalg.norm(got - want) / np.linalg.norm(want)
                key = (NV, dtype.__name__, 'inverse')
                worst[key] = max(worst.get(key, 0.0), err)
                assert ok
                xc, ok = block_solve_ldl(tiles, ycol, NV, dtype)
                got = from_cols(xc, NV).astype(np.float64)
                err = np.linalg.norm(got - want) / np.linalg.norm(want)
                key = (NV, dtype.__name__, 'ldl')
                worst[key] = max(worst.get(key, 0.0), err)
                assert ok
    for key in sorted(worst):
        print('NV=%d %-8s %-8s worst relative error %.3e' % (key[0], key[1], key[2], worst[key]))
    bad = [k for k, v in worst.items() if v > (1e-11 if k[1] == 'float64' else 2e-4)]
    sys.exit(1 if bad else 0)


if __name__ == '__main__' and len(sys.argv) == 1:
    main()


def ill_conditioned():
    """Rank-deficient Gram plus a small ridge (rows with fewer entries than dimensions and a
    small alpha): block solve against a plain fp32 Cholesky solve, errors against fp64."""
    rng = np.random.default_rng(7)
    NV, KP, rank = 4, 64, 50
    for n in (5, 20, 40, 80):
        errs_b, errs_c, errs_l = [], [], []
        for trial in range(5):
            F = np.zeros((KP, n))
            F[:rank] = rng.standard_normal((rank, n)) * 0.5 + 0.3
            A = F.dot(F.T)
            A[:rank, :rank] += 0.001 * n * np.eye(rank)
            A[rank:, rank:] += np.eye(KP - rank)
            b = F.dot(rng.integers(1, 6, n).astype(np.float64))
            want = np.linalg.solve(A, b)
            tiles, ycol = to_tiles(A.astype(np.float32), b.astype(np.float32), NV)
            xr, ok = block_solve(tiles, ycol, NV, np.float32)
            got = from_rows(xr, NV).astype(np.float64)
            errs_b.append(np.linalg.norm(got - want) / np.linalg.norm(want))
            xc_, ok = block_solve_ldl(tiles, ycol, NV, np.float32)
            got = from_cols(xc_, NV).astype(np.float64)
            errs_l.append(np.linalg.norm(got - want) / np.linalg.norm(want))
            A32, b32 = A.astype(np.float32), b.astype(np.float32)
            Lc = np.linalg.cholesky(A32)
            xc = np.linalg.solve(Lc.T, np.linalg.solve(Lc, b32)).astype(np.float64)
            errs_c.append(np.linalg.norm(xc - want) / np.linalg.norm(want))
        print('n=%3d cond=%.1e  block inverse %.2e  block LDL %.2e  fp32 cholesky %.2e' %
              (n, np.linalg.cond(A), max(errs_b), max(errs_l), max(errs_c)))


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'ill':
    ill_conditioned()
