"""Development aid: quotient-stage time for different gate mixes (which gate costs what)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, nlxpkg
nlx = nlxpkg.load()
ctx = nlx.Context(0)
for name, kw in [("all six", dict(pct_poseidon=30, pct_arithmetic=30, pct_base_sum=5, pct_constant=5)),
                 ("no poseidon", dict(pct_poseidon=0, pct_arithmetic=30, pct_base_sum=5, pct_constant=5)),
                 ("only const/pi/noop", dict(pct_poseidon=0, pct_arithmetic=0, pct_base_sum=0, pct_constant=5)),
                 ("all ten", dict(pct_poseidon=20, pct_arithmetic=20, pct_base_sum=5, pct_constant=5, pct_extension=20)),
                 ("no arithmetic/base", dict(pct_poseidon=30, pct_arithmetic=0, pct_base_sum=0, pct_constant=5)),
                 ("all nineteen", dict(pct_poseidon=10, pct_arithmetic=10, pct_base_sum=5, pct_constant=5, pct_extension=10, pct_misc=20, pct_u32=30))]:
    syn = nlx.SyntheticCircuit(16, seed=1, **kw)
    cd = nlx.CircuitData.from_synthetic(ctx, syn)
    w = torch.from_numpy(syn.wires.view(np.int64)).cuda()
    for _ in range(3):
        cd.prove_into(w, syn.public_inputs.ctypes.data)
    st = dict(cd.stage_times())
    print("%-20s quotient_eval %.3f ms  total %.2f ms" % (name, st["quotient_eval"], sum(st.values())))
    cd.close()
