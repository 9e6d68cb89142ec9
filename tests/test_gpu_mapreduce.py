"""GPU tests of the map-reduce dispatch with the real prover behind it (BASELINE.json configs[2]: the
single-tx VerifyCircuit shape = one map proof + the outer proof; and the 2x1 / 4x1 shapes): every proof of
the tree is accepted by the oracle verifier for its circuit AND byte-equal to what the oracle prover emits for the same
job, and the root digest does not depend on how many proofs are kept in flight."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class _Recording:
    """prove_fn wrapper keeping every proof with the circuit that produced it"""

    def __init__(self, prover):
        self.prover = prover
        self.proofs = []

    def __call__(self, kind, level, index, public_inputs):
        pr = self.prover(kind, level, index, public_inputs)
        self.proofs.append((kind, level if kind == "reduce" else 0, pr))
        return pr


@pytest.mark.parametrize("n_map", [1, 2, 4, 3])
def test_tree_proofs_verify_and_root_is_stable(nlx, ctx, orc, n_map):
    import torch
    mr = importlib.import_module("nlx_amd.mapreduce")
    plan = mr.TreePlan(n_map)
    assert plan.n_jobs == n_map + max(n_map - 1, 0) + 1
    prover = mr.GpuTreeProver(nlx, ctx, plan, 10, 9, torch=torch, workers=1)
    rec = _Recording(prover)
    root, stats = mr.run_tree(plan, rec)
    assert stats["proofs_by_this_rank"] == plan.n_jobs == len(rec.proofs)
    circs = prover.workers[0]["circ"]
    for kind, lvl, pr in rec.proofs:
        syn = circs[(kind, lvl)][0]
        oc = orc.Circuit.from_synthetic(syn)
        assert oc.verify(pr) == 1, (kind, lvl)
        oc.close()
    # the outer proof's public inputs are the digest of the root child's blob (output || proof), twice: the u64 count, then 8 words
    outer = rec.proofs[-1][2]
    assert int.from_bytes(outer[-72:-64], "little") == 8
    pis = np.frombuffer(outer[-64:], dtype=np.uint64)
    sio = importlib.import_module("nlx_amd.succinct_io")
    _, ids, batch = mr.default_request(n_map)
    want_out = sio.encode_verify_output([(i, True) for i in ids])
    assert stats["output"] == want_out and stats["outer_proof"] == outer
    child = mr.blob_digest(want_out, rec.proofs[-2][2])
    assert np.array_equal(pis[:4], child) and np.array_equal(pis[4:], child)
    assert np.array_equal(root, mr.blob_digest(want_out, outer))
    # the same tree proved by the ORACLE PROVER job by job (same circuits, same public inputs - a reduce job's inputs are the
    # digests of its children's blobs, so one differing byte anywhere below would change every proof above it): every one
    # of the tree's proofs is byte-equal, not only accepted
    oc = {key: orc.Circuit.from_synthetic(v[0]) for key, v in circs.items()}

    def oracle_prove(kind, level, index, public_inputs):
        syn = circs[(kind, level if kind == "reduce" else 0)][0]
        syn.set_public_inputs(public_inputs)
        return oc[(kind, level if kind == "reduce" else 0)].prove(syn.wires, syn.public_inputs)
    orec = _Recording(oracle_prove)
    oroot, ostats = mr.run_tree(plan, orec)
    assert [(k, l) for k, l, _ in orec.proofs] == [(k, l) for k, l, _ in rec.proofs]
    for (kind, lvl, want), (_, _, got) in zip(orec.proofs, rec.proofs):
        assert got == want, "tree proof (%s, level %d): GPU bytes differ from the oracle prover's" % (kind, lvl)
    assert np.array_equal(oroot, root) and ostats["output"] == want_out
    for c in oc.values():
        c.close()
    # same tree with three proofs in flight per level
    prover3 = mr.GpuTreeProver(nlx, ctx, plan, 10, 9, torch=torch, workers=3)
    root3, stats3 = mr.run_tree(plan, prover3)
    assert np.array_equal(root, root3) and stats3["output"] == want_out
