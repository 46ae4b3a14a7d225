import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from traversability_estimation_amd import capi
from oracle import oracle
from tests.test_gpu_random import draw_case
from tests.test_gpu_chain import both_fp
capi.load(); oracle.build()
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 6193
rows, cols, res, pos, elev, over = draw_case(seed)
got, want, op = both_fp(capi, oracle, elev, rows, cols, res, pos=pos, **over)
a, b = got["slope_footprint"], want["slope_footprint"]
bad = np.nonzero(np.isnan(a) != np.isnan(b))[0]
print("invalid fraction", float(np.isnan(elev).mean()), "bad cells", bad)
for k in bad:
    print(k, divmod(int(k), rows), "slope gpu", repr(float(got["traversability_slope"][k])), "oracle", repr(float(want["traversability_slope"][k])),
          "memo gpu", a[k], "oracle", b[k])
    # neighbourhood slopes that decide checkForSlope (zeros in circle 3 res)
    j, i = divmod(int(k), rows)
    g = got["traversability_slope"].reshape(cols, rows); w = want["traversability_slope"].reshape(cols, rows)
    sl = (slice(max(0, j - 4), j + 5), slice(max(0, i - 4), i + 5))
    print("zeros gpu", int((g[sl] == 0).sum()), "oracle", int((w[sl] == 0).sum()), "differing zero pattern", np.argwhere((g[sl] == 0) != (w[sl] == 0)).tolist())
    dd = np.argwhere((g[sl] == 0) != (w[sl] == 0))
    for (dj, di) in dd:
        print("   cell", (sl[0].start + dj, sl[1].start + di), "gpu", repr(float(g[sl][dj, di])), "oracle", repr(float(w[sl][dj, di])))
