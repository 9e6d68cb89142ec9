"""GPU tests of the map-reduce dispatch with the real prover behind it (BASELINE.json configs[2]: the
single-tx VerifyCircuit shape = one map proof + the outer proof; and the 2x1 / 4x1 shapes): every proof of
the tree is accepted by the oracle verifier for its circuit, and the root digest does not depend on how many
proofs are kept in flight."""
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
    # same tree with three proofs in flight per level
    prover3 = mr.GpuTreeProver(nlx, ctx, plan, 10, 9, torch=torch, workers=3)
    root3, stats3 = mr.run_tree(plan, prover3)
    assert np.array_equal(root, root3) and stats3["output"] == want_out
