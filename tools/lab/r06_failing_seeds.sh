# the seeds the round-6 sweep flagged, one by one with their assertion messages
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; O=$ROOT/gpurun_out/r06_sweep; mkdir -p $O; cd $ROOT
python - <<'PY' > $O/failing.txt 2>&1
import numpy as np, sys
from traversability_estimation_amd import capi
from oracle import oracle as O
from tests.test_gpu_random import draw_case
from tests.test_gpu_chain import both_fp
from tests.helpers import OUT_LAYERS, compare_layer
capi.load(); O.build()
for seed in [int(v) for v in __import__("os").environ.get("TE_SEEDS", "20477,20975,21402,21599,22405,22103,22633,22993,23804").split(",")]:
    rows, cols, res, pos, elev, over = draw_case(seed)
    if (over["fp_radius"] + over["fp_offset"]) / res > 19.5: over["fp_offset"] = 0.0
    got, want, op = both_fp(capi, O, elev, rows, cols, res, pos=pos, **over)
    print("seed", seed, rows, cols, res, {k: (round(v / res, 4) if "radius" in k else v) for k, v in over.items()})
    for k in list(OUT_LAYERS) + ["traversability_footprint"]:
        g, w = np.asarray(got[k]).reshape(cols, rows), np.asarray(want[k]).reshape(cols, rows)
        n_bad, mx, nn = compare_layer(k, g, w)
        if n_bad:
            bad = np.argwhere((np.isnan(g) != np.isnan(w)) | (np.abs(g.astype(np.float64) - w.astype(np.float64)) > 1e-5))
            print("   ", k, "mismatches", n_bad, "nan-pattern", nn, "max", mx, "first cells (j, i):", bad[:4].tolist(), "got", [float(g[tuple(b)]) for b in bad[:4]], "want", [float(w[tuple(b)]) for b in bad[:4]])
            b = bad[0]
            print("    elevation around:", elev.reshape(cols, rows)[max(0, b[0] - 1):b[0] + 2, max(0, b[1] - 1):b[1] + 2].tolist())
PY
cat $O/failing.txt
