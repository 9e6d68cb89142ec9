import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np, torch, nlxpkg
nlx = nlxpkg.load()
ctx = nlx.Context(0)
base=dict(pct_poseidon=20, pct_arithmetic=20, pct_base_sum=5, pct_constant=5)
for name, kw in [("six", {}), ("six+ext", dict(pct_extension=10)), ("six+misc", dict(pct_misc=10)), ("six+u32", dict(pct_u32=15)), ("all", dict(pct_extension=10,pct_misc=10,pct_u32=15))]:
    k=dict(base); k.update(kw)
    syn = nlx.SyntheticCircuit(16, seed=1, **k)
    cd = nlx.CircuitData.from_synthetic(ctx, syn)
    w = torch.from_numpy(syn.wires.view(np.int64)).cuda()
    for _ in range(3):
        cd.prove_into(w, syn.public_inputs.ctypes.data)
    st = dict(cd.stage_times())
    print("%-12s gates %2d sel %d quotient_eval %.3f ms" % (name, syn.num_gates, syn.num_selectors, st["quotient_eval"]))
    cd.close()
