"""Rehearsal of the wrap's quotient chain split over ranks (near-light-client_amd/plonk_split.py) under gloo:
   python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tests/tools/plonk_split_rehearsal.py [log_n]
The SCHEDULE is the product's; the three compute steps are the big-integer model's (oracle/bn254_py.py - test infrastructure),
restated per column / per slice of points, and rank 0 compares the result with the model's unsplit plonk_quotient.  Every rank
builds the same instance from one seed but hands the schedule only the columns it owns."""
import importlib.util
import os
import random
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    import torch
    import torch.distributed as dist
    import bn254_py as bn
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    pkg = types.ModuleType("nlx_units")
    pkg.__path__ = [os.path.join(ROOT, "near-light-client_amd")]
    sys.modules["nlx_units"] = pkg
    sys.modules["nlx_units._lib"] = types.SimpleNamespace(dll=None)   # the schedule needs no library

    def load(name):
        spec = importlib.util.spec_from_file_location("nlx_units." + name, os.path.join(ROOT, "near-light-client_amd", name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules["nlx_units." + name] = mod
        spec.loader.exec_module(mod)
        return mod
    load("split_ntt")
    ps = load("plonk_split")

    R = bn.R
    log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    n, n4 = 1 << log_n, 4 << log_n
    rng = random.Random(1234 + log_n)
    shift, k1, k2 = 5, 5, 25
    alpha, beta, gamma = (rng.randrange(R) for _ in range(3))
    p = bn.plonk_witness(log_n, rng, k1, k2, beta, gamma)
    names = ["ql", "qr", "qm", "qo", "qk", "s1", "s2", "s3", "l", "r", "o", "z"]

    def to_t(vals):
        return torch.tensor([[(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)] for v in vals], dtype=torch.uint64).view(torch.int64)

    def from_t(t):
        rows = t.view(torch.uint64).tolist()
        return [sum(int(x) << (64 * i) for i, x in enumerate(r)) for r in rows]

    def transform(name, values):                      # one column: values on H -> 4n coset evaluations
        return to_t(bn.coset_evals(values, shift))

    w4 = bn.root_of_unity(log_n + 2)

    def pointwise(cols, first, count):                # the model's per-point formula on points first .. first + count - 1
        ev = {k: from_t(v) for k, v in cols.items()}
        out, x = [], shift * pow(w4, first, R) % R
        for i in range(count):
            l, r, o, z, zn = ev["l"][i], ev["r"][i], ev["o"][i], ev["z"][i], ev["z"][i + 4]   # z's halo: four points ahead
            gate = (ev["ql"][i] * l + ev["qr"][i] * r + ev["qm"][i] * l * r + ev["qo"][i] * o + ev["qk"][i]) % R
            f = (l + beta * x + gamma) * (r + beta * k1 * x + gamma) % R * (o + beta * k2 * x + gamma) % R * z % R
            g = (l + beta * ev["s1"][i] + gamma) * (r + beta * ev["s2"][i] + gamma) % R * (o + beta * ev["s3"][i] + gamma) % R * zn % R
            zh = (pow(x, n, R) - 1) % R
            l1 = zh * pow(n * (x - 1) % R, R - 2, R) % R
            num = (gate + alpha * (f - g) + alpha * alpha % R * l1 % R * (z - 1)) % R
            out.append(num * pow(zh, R - 2, R) % R)
            x = x * w4 % R
        return to_t(out)

    def finish(t_evals):                              # rank 0: the one inverse transform, off the coset
        c = bn.ntt(from_t(t_evals), inverse=True)
        sinv, s, out = pow(shift, R - 2, R), 1, []
        for j in range(n4):
            out.append(c[j] * s % R)
            s = s * sinv % R
        return out

    local = {nm: p[nm] for j, nm in enumerate(names) if ps.column_owner(j, world) == rank}   # only what this rank owns
    got = ps.split_quotient_chain(names, local, n4, rank, world, dist, transform, pointwise, finish, device=torch.device("cpu"))
    if rank == 0:
        want = bn.plonk_quotient(p, shift, k1, k2, alpha, beta, gamma)
        assert got == want, "split chain differs from the model's"
        assert all(c == 0 for c in got[3 * n:]), "a satisfying witness: the top n coefficients vanish"
    else:
        assert got is None
    dist.barrier()
    print("plonk_split ok rank %d of %d, 2^%d gates, %d columns here" % (rank, world, log_n, len(local)), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
