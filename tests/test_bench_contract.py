"""bench.py prints ONE JSON line with the driver's contract fields (plus roofline and cpu_baseline); checked on a small
instance of the default workload and of the secondary ones."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config")


def run_bench(*args, timeout=300):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_parses_its_flags():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert r.returncode == 0 and "--gpus" in r.stdout and "--steps" in r.stdout and "--warmup" in r.stdout


@pytest.mark.gpu
def test_default_workload_line():
    d = run_bench("--gpus", "1", "--steps", "3", "--warmup", "1", "--log-n", "12")
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["scaling"] == "weak" and d["data"] == "synthetic" and "workload" in d["config"]
    assert d["value"] > 0 and abs(d["value"] * d["ms_per_step"] / 1e3 - 1.0) < 1e-6          # proofs/s x s/proof = 1 at N = 1
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and "traffic" in rf
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == d["unit"] and cb["sample"]


@pytest.mark.gpu
@pytest.mark.parametrize("args", [("--workload", "stark", "--log-n", "10"), ("--workload", "sha256", "--log-blocks", "4"),
                                  ("--workload", "sha512", "--log-blocks", "3"), ("--workload", "ed25519", "--log-slots", "8")])
def test_secondary_workload_lines(args):
    d = run_bench(*args, "--steps", "2", "--warmup", "1", "--no-cpu-baseline")
    for k in REQUIRED:
        assert k in d, k
    assert d["value"] > 0 and d["cpu_baseline"] is None and "workload" in d["config"]
