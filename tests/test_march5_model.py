"""CPU check of the te_march5.h scheme (traversability_estimation_amd/csrc/te_march5.h): the pass / slot / emit / prefetch-queue
arithmetic of the scatter march, modelled in numpy (tools/lab/march5_model.py), must reproduce the direct disc sum and disc
maximum for every strip layout.  It pins the index logic the HIP kernels unroll at compile time; the kernels themselves are
compared with the oracle in the -m gpu tests."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_march5_index_model():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "lab", "march5_model.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ok" in r.stdout
