"""Seed 133562 of the random sweep: one cell whose slope score differs by 1.39e-5 (tolerance 1e-5).  Prints the cell, both
normals as float32 and the double normal of its disc computed here (numpy eigh on the centred covariance)."""
import os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
from tests.test_gpu_random import draw_case
from tests.helpers import to_te_params
from traversability_estimation_amd import capi
from oracle import oracle

capi.load()
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 133562
rows, cols, res, pos, elev, over = draw_case(seed)
over["fp_offset"] = 0.0
op = oracle.default_params(**over)
g = oracle.geom(rows, cols, res, pos)
want = oracle.chain(g, op, elev, want_normals=True)
with capi.Context(0) as ctx:
    ctx.set_params(to_te_params(capi, op))
    ctx.set_geometry(rows, cols, 1, res, pos)
    ctx.upload_elevation(elev)
    ctx.run_chain(capi.RUN_KEEP_NORMALS)
    ctx.sync()
    got = {k: ctx.download(k).reshape(-1) for k in ("traversability_slope", "surface_normal_x", "surface_normal_y", "surface_normal_z")}
a, b = got["traversability_slope"], want["traversability_slope"].reshape(-1)
d = np.abs(a.astype(np.float64) - b)
for c in np.argsort(-np.nan_to_num(d))[:3]:
    i, j = int(c % rows), int(c // rows)
    print(f"cell {c} (i {i}, j {j}): slope here {a[c]!r} oracle {b[c]!r} diff {d[c]:.3e}")
    for k in ("surface_normal_x", "surface_normal_y", "surface_normal_z"):
        x, y = got[k][c], want[k].reshape(-1)[c]
        print(f"   {k}: here {x!r} ({x.view(np.uint32):#x}) oracle {y!r} ({np.float32(y).view(np.uint32):#x})")
    # the disc in double
    e = elev.reshape(cols, rows) if elev.ndim == 1 else elev
    R = over["normals_radius"]
    pts = []
    hw = int(R / res) + 1
    for dj in range(-hw, hw + 1):
        for di in range(-hw, hw + 1):
            ii, jj = i + di, j + dj
            if 0 <= ii < rows and 0 <= jj < cols and (di * res) ** 2 + (dj * res) ** 2 <= R * R and np.isfinite(e[jj, ii]):
                pts.append((-di * res, -dj * res, float(e[jj, ii])))
    P = np.array(pts, dtype=np.float64)
    C = np.cov(P.T, bias=True)
    w, V = np.linalg.eigh(C)
    n = V[:, 0] * np.sign(V[2, 0])
    nzf = np.float32(n[2])
    print(f"   disc of {len(pts)} cells: eigenvalues {w}, normal {n!r}")
    print(f"   nz in double {n[2]!r}: 1 - nz = {1 - n[2]:.6e}; float32 {nzf!r} ({nzf.view(np.uint32):#x}); distance to the rounding boundary in float32 ulps: "
          f"{abs((n[2] - float(nzf)) / np.spacing(nzf)) :.6f}")
    crit = over["slope_critical"]
    for v in (got["surface_normal_z"][c], want["surface_normal_z"].reshape(-1)[c]):
        print(f"   slope score from nz {v!r}: {1.0 - np.arccos(np.float64(v)) / crit!r}")
