# timeline of the default-parameter chain-only launch at res 0.03 (4096^2): do the step kernels run beside the normals kernel?
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; O=$ROOT/gpurun_out/r06g; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
cat > /tmp/one.py <<'PY'
import sys, numpy as np
sys.path.insert(0, "/root/repo")
from traversability_estimation_amd import capi, synth
capi.load()
n = 4096
with capi.Context(0) as c:
    c.set_params(capi.default_params()); c.set_geometry(n, n, 1, 0.03); c.upload_elevation(synth.perlin_elevation(n, n, seed=1234))
    s = c.time_chain_samples(0, warmup=5, iters=10); print("chain ms", float(np.median(s)))
PY
rocprofv3 --kernel-trace -d $O/tl -o t --output-format csv -- python /tmp/one.py > $O/tl.log 2>&1
python - <<PY
import csv, glob, re
for f in glob.glob("$O/tl/**/*kernel_trace.csv", recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    tail = rows[-10:]
    t0 = int(tail[0]["Start_Timestamp"])
    for r in tail:
        m = re.search(r"k_[a-z0-9_]+", r["Kernel_Name"])
        print("%-22s start %8.1f end %8.1f queue %s" % (m.group(0) if m else r["Kernel_Name"][:22], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, r.get("Queue_Id")))
PY
grep "chain ms" $O/tl.log
find $O -name "*kernel_trace.csv" -delete
