"""Shared helpers of the parity tests."""
import numpy as np

OUT_LAYERS = ("traversability_slope", "traversability_step", "traversability_roughness", "traversability")
TOL = 1e-5  # BASELINE.json north_star: outputs match the reference CPU filters within 1e-5 per cell


def compare_layer(name, got, want, tol=TOL):
    """NaN positions must match exactly; finite cells within tol.  Returns (n_mismatch, max_abs_err)."""
    got = np.asarray(got, np.float32).reshape(-1)
    want = np.asarray(want, np.float32).reshape(-1)
    assert got.shape == want.shape, (name, got.shape, want.shape)
    gn, wn = np.isnan(got), np.isnan(want)
    nan_mismatch = int((gn != wn).sum())
    both = ~gn & ~wn
    err = np.abs(got[both].astype(np.float64) - want[both].astype(np.float64))
    n_bad = int((err > tol).sum()) + nan_mismatch
    return n_bad, (float(err.max()) if err.size else 0.0), nan_mismatch


def assert_layers_match(got, want, layers=OUT_LAYERS, tol=TOL, ctx=""):
    report = []
    ok = True
    for k in layers:
        n_bad, mx, nn = compare_layer(k, got[k], want[k], tol)
        report.append(f"{k}: mismatches={n_bad} (nan-pattern {nn}) max|d|={mx:.3g}")
        ok &= (n_bad == 0)
    assert ok, f"{ctx}\n  " + "\n  ".join(report)
    return report


def orient_horizontal_normals(layers, nz_ref):
    """The SIGN of a horizontal normal is not defined by the filter: NormalVectorsFilter flips a normal only if its z
    component is negative, so where nz is 0 -- three collinear points, the usual disc of a one-cell tie radius whose cells
    across one axis were rejected -- the eigen-solver's rounding noise decides between n and -n (in the reference as in the
    oracle; slope, roughness and every later layer are the same for both).  Returns copies of surface_normal_x / _y with the
    horizontal normals (|nz_ref| <= 1e-6) turned so that their larger component is positive."""
    nx = np.array(layers["surface_normal_x"], np.float32).reshape(-1)
    ny = np.array(layers["surface_normal_y"], np.float32).reshape(-1)
    flat = np.abs(np.asarray(nz_ref, np.float32).reshape(-1)) <= 1e-6
    lead = np.where(np.abs(nx) >= np.abs(ny), nx, ny)
    sgn = np.where(flat & (lead < 0), -1.0, 1.0).astype(np.float32)
    out = dict(layers)
    out["surface_normal_x"], out["surface_normal_y"] = nx * sgn, ny * sgn
    return out


def to_te_params(capi, op):
    """oracle Params -> te_params (same field names)."""
    p = capi.default_params()
    for f, _ in op._fields_:
        setattr(p, f, getattr(op, f))
    return p
