"""CPU tests of the multi-GPU map-reduce dispatch (gloo, world size 2): every job runs exactly once,
the exchange is one all-gather per level, and the root digest does not depend on the rank count.
The prover behind the jobs is the CPU oracle at a tiny size (real proofs, real digests)."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT

N_MAP = 8


def _make_prover():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import nlxpkg
    import oracle_py
    nlx = nlxpkg.load()
    cache = {}

    def get(kind, level):
        key = (kind, level)
        if key not in cache:
            seed = {"map": 1, "reduce": 10 + level, "outer": 99}[kind]
            syn = nlx.SyntheticCircuit(5, seed=seed, num_public_inputs=8, pct_poseidon=10)
            cache[key] = (syn, oracle_py.Circuit.from_synthetic(syn))
        return cache[key]

    calls = []

    def prove_fn(kind, level, index, pis):
        syn, circ = get(kind, level)
        if pis is None:
            pis = np.array([(index * 7919 + k) % 65521 for k in range(8)], dtype=np.uint64)
        syn.set_public_inputs(pis)
        proof = circ.prove(syn.wires, syn.public_inputs)
        assert circ.verify(proof) == 1
        calls.append((kind, level, index))
        return proof
    return nlx, prove_fn, calls


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    nlx, prove_fn, calls = _make_prover()
    mr = sys.modules["nlx_amd"].mapreduce if hasattr(sys.modules["nlx_amd"], "mapreduce") else None
    if mr is None:
        import importlib
        mr = importlib.import_module("nlx_amd.mapreduce")
    plan = mr.TreePlan(N_MAP)
    root, stats = mr.run_tree(plan, prove_fn, rank, world, dist)
    q.put((rank, [int(x) for x in root], sorted(calls), stats["proofs_by_this_rank"]))
    dist.barrier()
    dist.destroy_process_group()


def _run(world, port):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(res)


def test_tree_plan():
    import importlib
    sys.path.insert(0, ROOT)
    import nlxpkg
    nlxpkg.load()
    mr = importlib.import_module("nlx_amd.mapreduce")
    plan = mr.TreePlan(32)
    assert plan.levels == [16, 8, 4, 2, 1] and plan.n_jobs == 32 + 31 + 1  # nearx verify.rs:69-90, 128 x 4
    with pytest.raises(ValueError):
        mr.TreePlan(12)


def test_root_independent_of_world_size():
    one = _run(1, 29611)
    two = _run(2, 29612)
    root1 = one[0][1]
    assert all(r[1] == root1 for r in two), "root digest depends on the number of ranks"
    jobs1 = one[0][2]
    jobs2 = sorted(two[0][2] + two[1][2])
    assert jobs1 == jobs2 and len(jobs1) == N_MAP + (N_MAP - 1) + 1  # every job exactly once
    # round-robin ownership: both ranks did real work
    assert two[0][3] > 0 and two[1][3] > 0 and two[0][3] + two[1][3] == len(jobs1)
