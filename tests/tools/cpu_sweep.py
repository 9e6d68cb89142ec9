"""CPU side of BASELINE.json configs[0] (plumbing, no GPU): the oracle prover (C port, OpenMP) on the default
nineteen-gate workload at several sizes.  Needs libnlx.so only for the workload generator (host code).
    python tests/tools/cpu_sweep.py [max_log_n]   ->  one JSON object per line"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
cores = min(len(os.sched_getaffinity(0)), 16)
os.environ["OMP_NUM_THREADS"] = str(cores)
import nlxpkg  # noqa: E402
import oracle_py  # noqa: E402

nlx = nlxpkg.load()
mix = dict(pct_poseidon=25, pct_arithmetic=20, pct_base_sum=5, pct_constant=5, pct_extension=10, pct_misc=10, pct_u32=15)
for log_n in range(10, (int(sys.argv[1]) if len(sys.argv) > 1 else 13) + 1):
    syn = nlx.SyntheticCircuit(log_n, seed=99, num_public_inputs=64, **mix)
    circ = oracle_py.Circuit.from_synthetic(syn)
    t = time.time()
    proof = circ.prove(syn.wires, syn.public_inputs)
    dt = time.time() - t
    ok = circ.verify(proof) == 1
    circ.close()
    print(json.dumps({"log_n": log_n, "cores": cores, "seconds": round(dt, 3), "proofs_per_s": round(1 / dt, 4),
                      "proof_bytes": len(proof), "verifier_accepts": ok}), flush=True)
