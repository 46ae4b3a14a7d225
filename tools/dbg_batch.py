import sys, numpy as np
sys.path.insert(0, '/root/repo')
from traversability_estimation_amd import capi, synth
rows, cols, res = 96, 80, 0.05
for B in [int(x) for x in sys.argv[1:]]:
    elevs = np.stack([synth.perlin_elevation(rows, cols, seed=2000 + b) for b in range(B)])
    r = synth.benchmark_radius(5, res)
    p = capi.default_params(normals_radius=r, rough_radius=r, step_radius1=r, step_radius2=r)
    with capi.Context(0) as ctx:
        ctx.set_params(p); ctx.set_geometry(rows, cols, B, res); ctx.upload_elevation(elevs)
        ctx.run_chain(0); ctx.sync()
        print('B', B, 'ok', np.nanmean(ctx.download('traversability')), flush=True)
