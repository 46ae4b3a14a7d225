"""bench.py's surface on the GPU box: every BASELINE configuration is ONE command (`--config cfgK`), the reference's parameter
files are read unchanged (`--yaml`), and each line carries the contract's keys, a roofline, and a parity check that passed.
(Short runs: the figures themselves are profiles/r06_configs.json's business.)"""
import json
import os
import subprocess
import sys

import pytest

from tests.conftest import ROOT

pytestmark = pytest.mark.gpu

KEYS = ("metric", "value", "unit", "n_gpus", "ranks", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
        "roofline", "parity_check")


def _bench(*args, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), env=e, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    for k in KEYS:
        assert k in line, k
    assert line["parity_check"]["ok"] and all(v["mismatches"] == 0 for v in line["parity_check"]["layers"].values())
    rf = line["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    return line


@pytest.mark.parametrize("cfg,cells,fp", [("cfg1", 100 * 133, False), ("cfg2", 1024 * 1024, False), ("cfg2", 1024 * 1024, True)])
def test_small_configurations_are_one_command(cfg, cells, fp):
    line = _bench("--config", cfg, "--steps", "20", "--warmup", "5", "--cpu-seconds", "2", "--no-cpu-all-cores", *(["--footprint"] if fp else []))
    assert line["config"]["name"] == cfg and line["config"]["map_cells"] == cells and line["config"]["footprint"] is fp
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["cores"] == 1 and line["cpu_baseline"]["value"] > 0
    assert abs(line["value"] - cells * 20 / (line["ms_per_step"] * 1e-3 * 20)) < 1e-6 * line["value"]
    assert ("traversability_footprint" in line["parity_check"]["layers"]) is fp


def test_streaming_configuration_counts_the_dirty_tile():
    line = _bench("--config", "cfg5", "--steps", "32", "--warmup", "8", "--no-cpu-baseline")
    assert line["config"]["map_cells"] == 8192 * 8192 and line["tick_mode"]["timed"] == "sync" and line["tick_mode"]["stream_ms_per_tick"] > 0
    assert abs(line["value"] - 256 * 256 * line["ticks_per_s"]) < 1e-6 * line["value"]
    assert len(line["parity_check"]["windows"]) == 4  # the windows around the last four tiles


def test_the_references_parameter_values_through_yaml(tmp_path):
    from tests.test_params_yaml import FOOTPRINT, SHIPPED
    f, fp = tmp_path / "robot_filter_parameter.yaml", tmp_path / "robot_footprint_parameter.yaml"
    f.write_text(SHIPPED)
    fp.write_text(FOOTPRINT)
    line = _bench("--yaml", str(f), "--footprint-yaml", str(fp), "--size", "512", "--steps", "20", "--warmup", "5", "--no-cpu-baseline")
    assert "robot_filter_parameter.yaml" in line["config"]["workload"] and abs(line["config"]["radius_cells"] - 1.0) < 1e-9  # 0.05 m on a 0.05 m map
    assert line["parity_check"]["layers"]["traversability_footprint"]["cells"] == 512 * 512
