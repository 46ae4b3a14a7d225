"""Which centres of a tie radius keep a symmetric disc?  (Round 6, the review's item 4b: "closed-form tail for centres whose
circle cells are all accepted, general tail otherwise, decided per lane".)

At a radius of exactly R cells the cells (+-R, 0) and (0, +-R) lie ON the circle and CircleIterator::isInside accepts each
of them from the rounded double positions of centre and cell (oracle/te_oracle.c: CIRCLE_FOREACH; te_geom.h: cell_x).  The
closed-form tail of the marching kernels needs a disc with si = sj = sij = 0 and sii = sjj: all four accepted or all four
rejected.  This script counts, for the bench geometry, the state of the (+-R, 0) pair per map row index (= per lane of the
march: a lane keeps its row index for the whole strip) and how often 64 consecutive lanes share one state.

    python tools/lab/tie_states.py [cells] [res] [R]
"""
import sys

import numpy as np


def pair_states(n, res, R, pos=0.0):
    length = n * res
    ax = pos + (0.5 * length - 0.5 * res)
    x = ax + res * (-np.arange(n, dtype=np.float64))
    r2 = (R * res) ** 2
    k = np.arange(n)

    def accepted(d):
        kk = np.clip(k + d, 0, n - 1)
        dx = x[kk] - x
        return np.where((k + d >= 0) & (k + d < n), dx * dx + 0.0 <= r2, True)

    ap, am = accepted(R), accepted(-R)
    return np.where(ap & am, 0, np.where(~ap & ~am, 1, 2))  # 0 both accepted, 1 both rejected, 2 one of them


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    res = float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
    R = int(sys.argv[3]) if len(sys.argv) > 3 else 9
    st = pair_states(n, res, R)
    frac = [float((st == s).mean()) for s in (0, 1, 2)]
    blocks = st[: n // 64 * 64].reshape(-1, 64)
    uniform = float((blocks == blocks[:, :1]).all(axis=1).mean())
    # a centre's disc is symmetric when its x pair and its y pair are in the SAME symmetric state; the y pair's state is the
    # same function of the column index
    sym = frac[0] * frac[0] + frac[1] * frac[1]
    print(f"{n} cells at {res} m, radius {R} cells: pair both accepted {frac[0]:.3f}, both rejected {frac[1]:.3f}, one of the two {frac[2]:.3f}")
    print(f"centres with a symmetric disc (x and y pairs in the same symmetric state): {sym:.3f}")
    print(f"blocks of 64 lanes that share one state: {uniform:.3f}; state changes along the axis: {int((np.diff(st) != 0).sum())}")
    print("states of lanes 1000..1063:", "".join(map(str, st[1000:1064])))


if __name__ == "__main__":
    main()
