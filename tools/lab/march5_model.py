#!/usr/bin/env python3
"""CPU model of te_march5.h's pass / slot / emit arithmetic (no GPU needed): a disc sum and a disc maximum computed by
the scatter march -- two rows per pass, 2R+1 rotating accumulators, body unrolled over P passes, early exit per pass,
prefetch queue of C passes with its rotation -- against the direct sums, for every shape and a few strip layouts."""
import math
import sys

import numpy as np


def isqrt(v):
    r = 0
    while (r + 1) * (r + 1) <= v:
        r += 1
    return r


def hw(Q, d):
    return isqrt(Q - d * d)


def direct(z, Q, op, ident):
    R = isqrt(Q)
    cols, rows = z.shape  # z[j, i]
    out = np.full_like(z, ident)
    for j in range(cols):
        for i in range(rows):
            acc = ident
            for dj in range(-R, R + 1):
                w = hw(Q, abs(dj))
                jj = j + dj
                if jj < 0 or jj >= cols:
                    continue
                for di in range(-w, w + 1):
                    ii = i + di
                    if 0 <= ii < rows:
                        acc = op(acc, z[jj, ii])
            out[j, i] = acc
    return out


def march(z, Q, op, ident, js, jend, C=2):
    """One strip: output rows [js, jend); all columns at once (a 'lane' per column, halo = direct indexing)."""
    R = isqrt(Q)
    P = 2 * R + 1
    cols, rows = z.shape
    out = {}

    def row(r):  # staged row: identity outside the map
        return z[r] if 0 <= r < cols else np.full(rows, ident, z.dtype)

    def build(v):  # nested run values S[d], d = 0..R
        S = [v.copy()]
        for d in range(1, R + 1):
            left = np.concatenate([np.full(d, ident, z.dtype), v[:-d]]) if d < rows else np.full(rows, ident, z.dtype)
            right = np.concatenate([v[d:], np.full(d, ident, z.dtype)]) if d < rows else np.full(rows, ident, z.dtype)
            S.append(op(op(S[d - 1], left), right))
        return S

    acc = [np.full(rows, ident, z.dtype) for _ in range(P)]
    r = js - R
    r_end = jend + R
    queue = [None] * C
    next_load = [r]

    def load_pair(rr, q):
        assert rr == next_load[0], (rr, next_load[0])  # loads are issued in row order
        queue[q] = (rr, row(rr), row(rr + 1))
        next_load[0] += 2

    for q in range(C):
        load_pair(r + 2 * q, q)
    while True:
        done = False
        for pc in range(P):
            b = 2 * pc
            q = pc % C
            if r >= r_end:
                done = True
                break
            rr, v0, v1 = queue[q]
            assert rr == r, (rr, r, pc)  # the queue slot holds the rows of this pass
            load_pair(r + 2 * C, q)
            s1, s2 = build(v0), build(v1)

            def emit(j, a):
                if js <= j < jend:
                    assert j not in out
                    out[j] = a.copy()

            if R == 0:
                acc[0] = s1[0].copy(); emit(r, acc[0]); acc[0] = s2[0].copy(); emit(r + 1, acc[0])
            else:
                for sl in range(P):
                    e0 = ((sl - b % P) % P + P) % P
                    e1 = e0 - P if e0 > R else e0
                    if e1 == -R:
                        acc[sl] = op(acc[sl], s1[hw(Q, R)])
                        emit(r - R, acc[sl])
                        acc[sl] = s2[hw(Q, R)].copy()
                    else:
                        w1, w2 = hw(Q, abs(e1)), hw(Q, abs(e1 - 1))
                        acc[sl] = op(op(acc[sl], s1[w1]), s2[w2])
                        if e1 - 1 == -R:
                            emit(r + 1 - R, acc[sl])
                            acc[sl] = np.full(rows, ident, z.dtype)
            r += 2
        if done:
            break
        n = P % C
        if n:
            queue[:] = [queue[(s + n) % C] for s in range(C)]
    return out


def main():
    rng = np.random.default_rng(3)
    shapes = [0, 1, 2, 4, 5, 8, 9, 10, 13, 25, 26, 81]
    bad = 0
    for Q in shapes:
        R = isqrt(Q)
        cols, rows = 3 * R + 23, 2 * R + 9
        z = rng.integers(0, 1000, size=(cols, rows)).astype(np.int64)
        for name, op, ident in (("sum", np.add, 0), ("max", np.maximum, -1)):
            want = direct(z, Q, op, ident)
            for strip in (cols, 7, 1, 2 * R + 1, 38):
                got = np.full_like(z, -7)
                js = 0
                while js < cols:
                    jend = min(js + strip, cols)
                    o = march(z, Q, op, ident, js, jend)
                    assert sorted(o) == list(range(js, jend)), (Q, strip, sorted(o)[:4])
                    for j, a in o.items():
                        got[j] = a
                    js = jend
                if not np.array_equal(got, want):
                    bad += 1
                    print("MISMATCH", Q, name, strip, int((got != want).sum()))
    print("march5 model:", "FAILED" if bad else "ok", "(%d shapes)" % len(shapes))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
