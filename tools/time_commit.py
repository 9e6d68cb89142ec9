"""Quick stage timing of the commit pipeline on the GPU (development aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nlxpkg
nlx = nlxpkg.load()
import torch
ctx = nlx.Context(0)
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n_cols = int(sys.argv[2]) if len(sys.argv) > 2 else 135
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
vals = torch.randint(0, 2**62, (n_cols, 1 << log_n), dtype=torch.int64, device="cuda")
for i in range(reps + 1):
    torch.cuda.synchronize()
    t = time.time()
    pb = nlx.PolynomialBatch.from_values(ctx, vals, 3, 4)
    dt = time.time() - t
    if i:
        print("commit %dx2^%d: %.3f ms" % (n_cols, log_n, dt * 1e3), flush=True)
    pb.close()
