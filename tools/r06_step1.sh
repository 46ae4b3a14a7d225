#!/bin/bash
# round 6, first record: the new bench configurations, the YAML path, the roctx ranges, the default-YAML shapes
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; O=$ROOT/gpurun_out/r06b; mkdir -p $O
export TMPDIR=/tmp; cd $ROOT
python bench.py --config cfg1 --steps 200 --warmup 20 > $O/cfg1.json 2> $O/cfg1.err
python bench.py --config cfg1 --footprint --steps 200 --warmup 20 --no-cpu-all-cores > $O/cfg1_fp.json 2> $O/cfg1_fp.err
python bench.py --config cfg2 --steps 100 --warmup 20 > $O/cfg2.json 2> $O/cfg2.err
python bench.py --config cfg2 --footprint --steps 100 --warmup 20 --no-cpu-baseline > $O/cfg2_fp.json 2> $O/cfg2_fp.err
python bench.py --config cfg5 --steps 64 --warmup 8 --no-cpu-baseline > $O/cfg5.json 2> $O/cfg5.err
python -c "from tests.test_params_yaml import SHIPPED, FOOTPRINT; open('/tmp/f.yaml','w').write(SHIPPED); open('/tmp/fp.yaml','w').write(FOOTPRINT)"
python bench.py --yaml /tmp/f.yaml --footprint-yaml /tmp/fp.yaml --size 1024 --steps 50 --warmup 10 --no-cpu-all-cores > $O/yaml_1024.json 2> $O/yaml_1024.err
cd /tmp
rocprofv3 --marker-trace --kernel-trace --stats -d $O/marker -o m --output-format csv -- python $ROOT/bench.py --config cfg2 --footprint --steps 5 --warmup 2 --no-cpu-baseline --no-check > $O/marker.log 2>&1
find $O/marker -name "*kernel_trace.csv" -delete
python $ROOT/tools/defaults_bench.py > $O/defaults.json 2> $O/defaults.err
rocprofv3 --kernel-trace --stats -d $O/defaults_kt -o d --output-format csv -- python $ROOT/tools/defaults_bench.py profile > $O/defaults_kt.log 2>&1
find $O/defaults_kt -name "*kernel_trace.csv" -delete
for f in cfg1 cfg1_fp cfg2 cfg2_fp cfg5 yaml_1024; do echo "== $f"; cut -c1-400 $O/$f.json; tail -3 $O/$f.err; done
ls -R $O/marker | head -20
