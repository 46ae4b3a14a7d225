"""the default-parameter chain at res 0.03 (4096^2) in a process of its own, and after a res-0.05 context: does what ran before matter?"""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from traversability_estimation_amd import capi, synth
capi.load()
n = 4096
e = synth.perlin_elevation(n, n, seed=1234)
def run(res, tag):
    with capi.Context(0) as c:
        c.set_params(capi.default_params()); c.set_geometry(n, n, 1, res); c.upload_elevation(e)
        a = float(np.median(c.time_chain_samples(0, warmup=5, iters=50)))
        b = float(np.median(c.time_chain_samples(capi.RUN_SEQUENTIAL, warmup=5, iters=50)))
        print(tag, "res", res, "chain", round(a, 4), "sequential", round(b, 4))
order = sys.argv[1:] or ["0.03"]
for k, r in enumerate(order):
    run(float(r), f"#{k}")
