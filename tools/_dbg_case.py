import sys, numpy as np
sys.path.insert(0, '/root/repo')
from tests.test_gpu_random import draw_case
from tests.test_gpu_chain import both_fp
from oracle import oracle as O
from traversability_estimation_amd import capi
capi.load()
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1271
rows, cols, res, pos, elev, over = draw_case(seed)
got, want, op = both_fp(capi, O, elev, rows, cols, res, pos=pos, **over)
g = got['step_footprint'].reshape(cols, rows); w = want['step_footprint'].reshape(cols, rows)
bad = np.argwhere(~((g == w) | (np.isnan(g) & np.isnan(w))))
print('n bad', len(bad), 'crit_step', op.fp_critical_step if hasattr(op,'fp_critical_step') else None, 'max_gap', getattr(op,'fp_max_gap',None))
E = elev.reshape(cols, rows); S = want['traversability_step'].reshape(cols, rows)
for (j, i) in bad[:6]:
    print('cell i=%d j=%d gpu=%s oracle=%s elev=%.4f step=%s' % (i, j, g[j, i], w[j, i], E[j, i], S[j, i]))
    j0, j1, i0, i1 = max(0, j-3), min(cols, j+4), max(0, i-3), min(rows, i+4)
    print(np.array2string(E[j0:j1, i0:i1], precision=3, suppress_small=True))
    print(np.array2string(S[j0:j1, i0:i1], precision=2, suppress_small=True))
