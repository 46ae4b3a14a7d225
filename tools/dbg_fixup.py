"""Measurement aid: cells the normals march leaves to the fix-up pass (run with TE_DEBUG_SKIP_FIXUP=1)."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from traversability_estimation_amd import capi
capi.load()
d = np.load(os.path.join(ROOT, "tests", "golden", "bag_map.npz"))
rows, cols = int(d["rows"]), int(d["cols"])
with capi.Context(0) as c:
    c.set_params(capi.default_params())
    c.set_geometry(rows, cols, 1, float(d["resolution"]), tuple(d["position"]))
    c.upload_elevation(d["elevation"])
    c.run_chain(0)
    c.sync()
    for layer in ("traversability_slope", "traversability_roughness"):
        s = c.download(layer)
        e = np.asarray(d["elevation"], np.float32).reshape(-1)
        todo = np.isnan(s) & np.isfinite(e)
        print(os.environ.get("TRAVGPU_LIB", "")[-8:], layer, "bag: cells left to the fix-up", int(todo.sum()), "of", int(np.isfinite(e).sum()), "valid")
        t = todo.reshape(cols, rows)
        print("  per map row (j):", "".join("%x" % min(15, int(v)) for v in t.sum(axis=1)))
        print("  per column (i):", "".join("%x" % min(15, int(v)) for v in t.sum(axis=0)))
