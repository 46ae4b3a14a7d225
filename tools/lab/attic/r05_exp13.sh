cd $GRAFT_REPO_ROOT
for k in 1 2; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > /tmp/b$k.json 2> /tmp/b$k.err
  tail -3 /tmp/b$k.err
  python - <<PY
import json
d = json.loads(open("/tmp/b$k.json").read().strip().splitlines()[-1])
print(round(d["ms_per_step"], 4), {k: v for k, v in d["host_path"].items() if "three" in k and "what" not in k})
PY
done
