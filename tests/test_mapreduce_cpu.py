"""CPU tests of the multi-GPU map-reduce dispatch (gloo, world size 2): every job runs exactly once, the exchange is
one all-gather per level of the children's (public output || proof) blobs, and neither the root digest nor the job's
output (VERIFY_AMT x 33 bytes, every id in request order) depends on the rank count.
The prover behind the jobs is the CPU oracle at a tiny size (real proofs, real blobs)."""
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
        syn.set_public_inputs(pis)
        proof = circ.prove(syn.wires, syn.public_inputs)
        assert circ.verify(proof) == 1
        calls.append((kind, level, index))
        return proof
    return nlx, prove_fn, calls


def _worker(rank, world, port, q, n_map=N_MAP):
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
    plan = mr.TreePlan(n_map)
    root, stats = mr.run_tree(plan, prove_fn, rank, world, dist)
    q.put((rank, [int(x) for x in root], sorted(calls), stats["proofs_by_this_rank"], stats["output"], stats["bytes_gathered"]))
    dist.barrier()
    dist.destroy_process_group()


def _run(world, port, n_map=N_MAP):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, n_map)) for r in range(world)]
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
    assert not any(plan.carried)
    # not a power of two: the unpaired node of a level moves up unchanged (12 -> 6 -> 3 -> 1 + carried -> 1)
    odd = mr.TreePlan(12)
    assert odd.levels == [6, 3, 1, 1] and odd.carried == [False, False, True, False] and odd.n_jobs == 12 + 11 + 1
    assert mr.TreePlan(5).levels == [2, 1, 1] and mr.TreePlan(1).levels == []
    with pytest.raises(ValueError):
        mr.TreePlan(0)


def test_blob_codec_and_digest():
    import importlib
    sys.path.insert(0, ROOT)
    import nlxpkg
    nlxpkg.load()
    mr = importlib.import_module("nlx_amd.mapreduce")
    b = mr.Blob(b"\x01" * 66, bytes(range(200)))
    again = mr.Blob.unpack(b.pack() + b"\0" * 40)       # padded to the level's widest blob on the wire
    assert again.output == b.output and again.proof == b.proof
    d = b.digest()
    assert d.shape == (4,) and all(int(x) < 0xFFFFFFFF00000001 for x in d)
    assert not np.array_equal(d, mr.Blob(b.output, b.proof[:-1] + b"\0").digest())   # the proof is bound
    assert not np.array_equal(d, mr.Blob(b"\x02" + b.output[1:], b.proof).digest())   # and so is the output
    with pytest.raises(ValueError):
        mr.Blob.unpack(b.pack()[:-1])


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
    # the job's output: every requested id, in request order, verified - the same bytes on every rank and rank count
    import importlib
    mr = importlib.import_module("nlx_amd.mapreduce")
    sio = importlib.import_module("nlx_amd.succinct_io")
    _, ids, batch = mr.default_request(N_MAP)
    want = sio.encode_verify_output([(i, True) for i in ids])
    assert len(want) == N_MAP * batch * 33
    assert one[0][4] == want and all(r[4] == want for r in two)
    # what crossed the wire per job is its whole blob (proof + output), not a 32-byte digest
    assert two[0][5] == one[0][5] > (N_MAP + N_MAP - 1) * 50_000


def test_odd_job_count_two_ranks():
    """5 map jobs (the reference asserts a power of two, nearx/src/main.rs:19; the tree here carries the unpaired node up)"""
    one = _run(1, 29613, 5)
    two = _run(2, 29614, 5)
    assert all(r[1] == one[0][1] for r in two) and all(r[4] == one[0][4] for r in two)
    assert len(one[0][2]) == 5 + 4 + 1 and sorted(two[0][2] + two[1][2]) == one[0][2]
