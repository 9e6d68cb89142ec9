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


@pytest.mark.gpu
@pytest.mark.parametrize("args", [("--log-n", "12"), ("--workload", "verify128", "--log-n", "10", "--map-log-n", "11")])
def test_driver_launch_line_two_ranks(args):
    """The driver's N > 1 command (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...`) with N = 2 on the one-GPU box: NLX_BENCH_REHEARSAL=1 puts both ranks on
    GPU 0 over gloo (RCCL refuses two ranks on one device), everything else is the code path of a real two-GPU run:
    rank / world from the environment, barrier, MAX over ranks, one JSON line from rank 0."""
    env = dict(os.environ, NLX_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29641", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", *args]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["value"] > 0 and "rehearsal" in d["config"]
    assert d["scaling"] == ("strong" if "verify128" in args else "weak")
