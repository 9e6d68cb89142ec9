"""Differential fuzz (run on the GPU box): random gate mixes / sizes / public-input counts; GPU proof bytes must
equal the oracle prover and the oracle verifier must accept.  `python tests/tools/fuzz_plonk.py [cases [max_log_n [seed]]]`
(max_log_n > 12 also exercises the multi-chunk opening kernels)."""
import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, _p)
import nlxpkg; nlx=nlxpkg.load()
import oracle_py as orc
import numpy as np
ctx=nlx.Context(0)
MAX_LOG_N=int(sys.argv[2]) if len(sys.argv) > 2 else 10
SEED=int(sys.argv[3]) if len(sys.argv) > 3 else 77
rng=np.random.default_rng(SEED)
bad=0; n=0; t0=time.time()
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    log_n=int(rng.integers(5,MAX_LOG_N+1))
    pct=[int(x) for x in rng.integers(0,30,7)]
    tot=sum(pct)
    if tot>95: pct=[p*90//tot for p in pct]
    kw=dict(pct_poseidon=pct[0],pct_arithmetic=pct[1],pct_base_sum=pct[2],pct_constant=max(pct[3],1),pct_extension=pct[4],pct_misc=pct[5],pct_u32=pct[6])
    syn=nlx.SyntheticCircuit(log_n, seed=5000+SEED+it, num_public_inputs=int(rng.integers(0,9)), **kw)
    ref=orc.Circuit.from_synthetic(syn); cd=nlx.CircuitData.from_synthetic(ctx, syn)
    want=ref.prove(syn.wires, syn.public_inputs); got=cd.prove(syn.wires, syn.public_inputs)
    ok = got==want and ref.verify(got)==1
    n+=1
    if not ok:
        bad+=1; print("MISMATCH", it, log_n, kw)
    cd.close(); ref.close()
print("fuzz: %d circuits, %d mismatches, %.1fs" % (n,bad,time.time()-t0))
