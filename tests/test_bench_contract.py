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


def test_gpus_flag_disagreeing_with_world_size_is_an_error():
    """--gpus 4 under WORLD_SIZE=2 must not measure something else and call it a 4-GPU run: exit code 2, before any GPU call"""
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True, timeout=120, cwd=ROOT, env=env)
    assert r.returncode == 2 and "WORLD_SIZE=2" in r.stderr and r.stdout.strip() == ""


def test_gpus_flag_without_launcher_starts_the_ranks_itself():
    """--gpus 2 with WORLD_SIZE unset: bench.py starts python -m torch.distributed.run ... itself (as a child, before any GPU
    call).  On a box without two GPUs the ranks fail, and so does the parent - it never falls back to one GPU."""
    import torch
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["NLX_BENCH_REHEARSAL"] = "0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--log-n", "10",
                        "--no-extra", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert "without a launcher, starting -m torch.distributed.run --nnodes=1 --nproc-per-node 2" in r.stderr
    if torch.cuda.device_count() < 2:
        assert r.returncode != 0 and not any(ln.startswith("{") for ln in r.stdout.splitlines())


@pytest.mark.gpu
def test_default_workload_line():
    """the default job = one full Sync proof (three STARKs of the mainnet step + the outer plonky2 proof, here at 2^12 rows)"""
    d = run_bench("--gpus", "1", "--steps", "3", "--warmup", "1", "--log-n", "12", timeout=600)
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["scaling"] == "weak" and d["data"].startswith("synthetic") and "workload" in d["config"]
    for part in ("SHA-256 STARK", "SHA-512 STARK", "Ed25519 STARK", "outer plonky2 proof"):
        assert part in d["config"]["workload"], part
    assert d["value"] > 0 and abs(d["value"] * d["ms_per_step"] / 1e3 - 1.0) < 1e-6          # proofs/s x s/proof = 1 at N = 1
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and "traffic" in rf
    rv = d["roofline_valu"]
    assert rv["unit"] == "Gperm/s" and 0 < rv["frac"] < 1.2 and abs(rv["frac"] - rv["achieved"] / rv["peak"]) < 1e-12
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == d["unit"] and cb["sample"]
    # the cpu_baseline leg proves the same inputs on both sides and compares the BYTES
    pc = d["parity_checked"]
    assert pc["all_bytes_equal"] is True and pc["outer"]["bytes_equal"] and pc["sha256"]["bytes_equal"] and pc["sha512"]["bytes_equal"]
    assert pc["ed25519"]["bytes_equal"] and pc["outer"]["oracle_verifier_accepts"]
    assert d["config"]["outer_rows_floor_from_stark_verification"]["total"] > 1 << 16
    # the headline runs the reference's STARK protocol: whole-row hash_or_noop leaves, every opening observed
    assert d["config"]["stark_variant"] == "starky" and d["config"]["leaf_group_cols"] == 0 and d["config"]["openings_group"] == 0


@pytest.mark.gpu
def test_outer_only_workload_line():
    d = run_bench("--workload", "outer", "--steps", "3", "--warmup", "1", "--log-n", "12")
    for k in REQUIRED:
        assert k in d, k
    assert d["roofline"]["kernel"] == "k_hash_lde_leaves" and d["cpu_baseline"]["value"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("args", [("--workload", "stark", "--log-n", "10"), ("--workload", "sha256", "--log-blocks", "4"),
                                  ("--workload", "sha512", "--log-blocks", "3"), ("--workload", "ed25519", "--log-slots", "8"),
                                  ("--workload", "ntt24", "--ntt-log-n", "18", "--ntt-cols", "4"), ("--workload", "msm24", "--ntt-log-n", "14"),
                                  ("--workload", "msm24", "--msm-group", "g2", "--ntt-log-n", "12"),
                                  ("--workload", "ntt24", "--ntt-field", "bn254", "--ntt-order", "dit", "--ntt-log-n", "16", "--ntt-cols", "2")])
def test_secondary_workload_lines(args):
    d = run_bench(*args, "--steps", "2", "--warmup", "1", "--no-cpu-baseline")
    for k in REQUIRED:
        assert k in d, k
    assert d["value"] > 0 and d["cpu_baseline"] is None and "workload" in d["config"]


def _launch_two_ranks(args, env_extra, port):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", *args]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["value"] > 0
    return d


VERIFY_ARGS = ("--workload", "verify128", "--reduce-log-n", "10", "--map-log-n", "11")


@pytest.mark.gpu
@pytest.mark.parametrize("args", [("--log-n", "12", "--no-extra"), VERIFY_ARGS])
def test_driver_launch_line_two_ranks(args):
    """The driver's N > 1 command (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...`) with N = 2 on the one-GPU box: NLX_BENCH_REHEARSAL=1 puts both ranks on
    GPU 0 over gloo (RCCL refuses two ranks on one device), everything else is the code path of a real two-GPU run:
    rank / world from the environment, barrier, MAX over ranks, one JSON line from rank 0.  The Verify job's root digest
    and output must equal the one-rank run's."""
    d = _launch_two_ranks(args, {"NLX_BENCH_REHEARSAL": "1"}, 29641)
    assert "rehearsal" in d["config"]
    assert d["scaling"] == ("strong" if "verify128" in args else "weak")
    if "verify128" in args:
        one = run_bench(*args, "--steps", "1", "--warmup", "0", "--no-cpu-baseline", timeout=600)
        assert one["config"]["root_digest"] == d["config"]["root_digest"], "root(ws=2) != root(ws=1)"
        assert d["config"]["output_lists_every_id_as_verified"] is True and one["config"]["output_lists_every_id_as_verified"] is True
        assert d["config"]["bytes_gathered_last_step"] == one["config"]["bytes_gathered_last_step"] > 64 * 50_000


@pytest.mark.gpu
def test_msm_split_over_two_ranks_joins_to_the_one_rank_result():
    """--workload msm24 with the points split over two ranks (rehearsal: both on GPU 0 over gloo): the partial results, joined
    by nlx_bn254_g1_sum on rank 0, are the one-rank result"""
    args = ("--workload", "msm24", "--ntt-log-n", "14")
    two = _launch_two_ranks(args, {"NLX_BENCH_REHEARSAL": "1"}, 29643)
    one = run_bench(*args, "--steps", "1", "--warmup", "0", "--no-cpu-baseline")
    assert two["config"]["points_per_rank"] * 2 == one["config"]["points_per_rank"] == 1 << 14
    assert two["config"]["result_sha256"] == one["config"]["result_sha256"]


@pytest.mark.gpu
def test_one_ntt_split_over_two_ranks_equals_the_one_rank_transform():
    """--workload ntt24 --ntt-split (BASELINE.json configs[4] "split over GPUs"): every 2^16-point transform split over two ranks
    (rehearsal: both on GPU 0, gloo) - one slice exchange, one cross-rank butterfly level, a 2^15-point transform per rank -
    reassembles to the one-rank result"""
    args = ("--workload", "ntt24", "--ntt-log-n", "16", "--ntt-cols", "3")
    two = _launch_two_ranks(args + ("--ntt-split",), {"NLX_BENCH_REHEARSAL": "1"}, 29644)
    one = run_bench(*args, "--steps", "1", "--warmup", "0", "--no-cpu-baseline")
    assert two["scaling"] == "strong" and two["config"]["bytes_sent_per_rank_per_step"] == 3 * (1 << 15) * 8
    assert two["config"]["result_sha256"] == one["config"]["result_sha256"] and one["config"]["result_sha256"] is not None


@pytest.mark.gpu
def test_default_line_carries_the_verify_record_on_two_ranks():
    """The driver's N > 1 command WITHOUT --no-extra (small sizes): after the Sync replicas every rank runs the 128 x 4 Verify job
    and rank 0's line carries its record - the strong-scaling figure of BASELINE.json configs[3] (gloo rehearsal on one GPU)."""
    d = _launch_two_ranks(("--log-n", "12", "--map-log-n", "11", "--reduce-log-n", "10"), {"NLX_BENCH_REHEARSAL": "1"}, 29644)
    v = d["verify128"]
    assert "error" not in v, v
    assert v["n_gpus"] == 2 and v["scaling"] == "strong" and v["output_ok"] is True and v["proofs_per_s"] > 0
    # the recorded job includes the map jobs' SHA-256 STARKs (one per rank over the jobs it owns), first entry of level_ms
    assert v["map_starks"] is True and v["map_starks_ms_per_job"] > 0 and v["level_ms"][0][0].startswith("map_starks_sha256")
    assert len(v["level_ms"]) == 8 and v["roofline"]["launches"] > 0
    one = run_bench("--workload", "verify128", "--map-log-n", "11", "--reduce-log-n", "10", "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
                    timeout=600)
    assert one["config"]["root_digest"] == v["root_digest"]


@pytest.mark.gpu
def test_two_ranks_over_rccl():
    """the real N = 2 path (backend nccl = RCCL, one rank per GPU): runs where two GPUs are visible, skipped on a one-GPU box"""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    d = _launch_two_ranks(VERIFY_ARGS, {}, 29642)
    assert "rehearsal" not in d["config"] and d["scaling"] == "strong"
    one = run_bench(*VERIFY_ARGS, "--steps", "1", "--warmup", "0", "--no-cpu-baseline", timeout=600)
    assert one["config"]["root_digest"] == d["config"]["root_digest"]


def _run_dist_units(backend, port, n=2):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "tools", "dist_units.py"), backend]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    assert r.stdout.count("dist_units ok") == n, r.stdout[-2000:]


def test_exchange_primitives_two_and_three_ranks_gloo():
    """mapreduce.all_gather_blobs (ragged blobs, job counts that are not multiples of the rank count, a rank that owns nothing)
    and split_ntt._exchange under torch.distributed.run on host tensors - the same script the RCCL test below runs"""
    _run_dist_units("gloo", 29651, 2)
    _run_dist_units("gloo", 29652, 3)


@pytest.mark.gpu
def test_exchange_primitives_two_ranks_rccl():
    """the same two primitives over backend nccl = RCCL with one rank per GPU and every tensor on the rank's own device: the
    first thing to run on a multi-GPU node (skipped where fewer than two GPUs are visible - RCCL refuses two ranks on one)"""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    _run_dist_units("nccl", 29653, 2)


@pytest.mark.parametrize("n", [2, 4])
def test_wrap_quotient_chain_split_schedule_under_gloo(n):
    """the PLONK quotient chain split over ranks - transforms by column, one all-to-all of point slices
    (z with its four-point halo), pointwise by points, the inverse transform on rank 0 - gives the unsplit chain's coefficients.
    The schedule is near-light-client_amd/plonk_split.py; the arithmetic in this rehearsal is the big-integer model's."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(29660 + n), os.path.join(ROOT, "tests", "tools", "plonk_split_rehearsal.py"), "4"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.count("plonk_split ok") == n, r.stdout[-2000:]
