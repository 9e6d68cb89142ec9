"""Wall time of a whole PLONK proof over BN254 on the device (near-light-client_amd/bn254_plonk.py) at 2^k gates.  The instance is the
cheapest satisfying one that still exercises every kernel at full size: all selectors zero, random wires, the identity
permutation (z = 1), an SRS of distinct points (i G: the pipeline does not care that it is not a power series).
  python3 tools/plonk_prove_timing.py 16 18"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import nlxpkg

nlx = nlxpkg.load()
import torch

P = nlx.bn254_plonk
R = P.R
ctx = nlx.Context(0)
for log_n in [int(a) for a in sys.argv[1:]] or [16]:
    n = 1 << log_n
    rng = np.random.default_rng(log_n)
    w = P.root_of_unity(log_n)
    ident, x = [], 1
    for _ in range(n):
        ident.append(x)
        x = x * w % R
    vals = {k: [0] * n for k in ("ql", "qr", "qm", "qo", "qk")}
    vals.update(s1=ident, s2=[5 * v % R for v in ident], s3=[25 * v % R for v in ident])
    srs = nlx.bn254_g1_multiples(ctx, (1, 2), n, device="cuda:0")
    pk = P.ProvingKey(ctx, vals, srs, 5, 25)
    # witness as fr.Element words resident in HBM (any words below r are Montgomery forms of some field elements)
    wires = [torch.from_numpy(np.stack([rng.integers(0, 2 ** 62, n), rng.integers(0, 2 ** 62, n), rng.integers(0, 2 ** 62, n),
                                        rng.integers(0, 2 ** 60, n)], axis=1).astype(np.int64)).cuda() for _ in range(3)]
    P.prove(pk, *wires)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        proof = P.prove(pk, *wires)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print("2^%d gates: %.1f ms per proof (witness resident in HBM; 9 MSMs of n points, 5 + 12 + 1 transforms of n / 4n, the grand product, "
          "two openings; proof = 9 points + 6 scalars)" % (log_n, dt * 1e3), flush=True)
