"""Multi-GPU dispatch of the VerifyCircuit map-reduce proof tree.

Mirrors plonky2x `frontend::mapreduce::generator::MapReduceDynamicGenerator` +
`backend::prover::LocalProver::batch_prove` as used by nearx (`nearx/src/verify.rs:69-90`,
registration at `:112-122`): N/B independent MAP proofs, then a binary tree of REDUCE proofs, then the
outer proof.  The reference runs them one after another on CPU ("No parallelisation", README.md:123);
here the jobs of every level are dealt round-robin over the ranks (one process per GPU) and the only
exchange is ONE all-gather per level of the children's digests (RCCL over xGMI on GPUs, gloo in the
CPU tests).  Results do not depend on the number of ranks.

The prover behind a job is injected (`prove_fn`) so the sharding logic is testable without a GPU.
"""
import numpy as np

P = 0xFFFFFFFF00000001


def proof_digest(proof_bytes):
    """4 field elements identifying a child proof: its first Merkle-cap entry (wires cap[0])."""
    return np.frombuffer(proof_bytes[:32], dtype=np.uint64).copy()


class TreePlan:
    """Static job list for `n_map` map proofs reduced pairwise to one (n_map a power of two)."""

    def __init__(self, n_map):
        if n_map < 1 or n_map & (n_map - 1):
            raise ValueError("n_map must be a power of two")
        self.n_map = n_map
        self.levels = []  # levels[l] = number of reduce jobs at reduce level l
        k = n_map
        while k > 1:
            k //= 2
            self.levels.append(k)

    @property
    def n_jobs(self):
        return self.n_map + sum(self.levels) + 1  # + outer proof


def owner(job_index, world):
    return job_index % world


def all_gather_digests(local, n_jobs, rank, world, dist, device=None):
    """local: dict job_index -> (4,) uint64.  Returns (n_jobs, 4) uint64 with every job's digest.
    One collective per level; payload is n_jobs * 32 bytes."""
    per = (n_jobs + world - 1) // world
    if dist is None or world == 1:
        out = np.zeros((n_jobs, 4), dtype=np.uint64)
        for j, d in local.items():
            out[j] = d
        return out
    import torch
    send = torch.zeros((per, 4), dtype=torch.int64, device=device)
    for j, d in local.items():
        send[j // world] = torch.from_numpy(np.asarray(d, dtype=np.uint64).view(np.int64)).to(send.device)
    recv = [torch.zeros_like(send) for _ in range(world)]
    dist.all_gather(recv, send)
    out = np.zeros((n_jobs, 4), dtype=np.uint64)
    for r in range(world):
        got = recv[r].cpu().numpy().view(np.uint64)
        for slot in range(per):
            j = slot * world + r
            if j < n_jobs:
                out[j] = got[slot]
    return out


def run_tree(plan, prove_fn, rank=0, world=1, dist=None, device=None):
    """Executes the whole tree.  prove_fn(kind, level, index, public_inputs) -> proof bytes, where kind
    is "map" | "reduce" | "outer"; map jobs get no children (public_inputs=None -> the job's own).
    Returns (root_digest, stats) on every rank; stats counts the proofs this rank produced."""
    done = 0
    # ---- map level ----
    local = {}
    for j in range(plan.n_map):
        if owner(j, world) == rank:
            local[j] = proof_digest(prove_fn("map", 0, j, None))
            done += 1
    digests = all_gather_digests(local, plan.n_map, rank, world, dist, device)
    # ---- reduce levels ----
    for lvl, n_jobs in enumerate(plan.levels):
        local = {}
        for j in range(n_jobs):
            if owner(j, world) == rank:
                pis = np.concatenate([digests[2 * j], digests[2 * j + 1]])
                local[j] = proof_digest(prove_fn("reduce", lvl, j, pis))
                done += 1
        digests = all_gather_digests(local, n_jobs, rank, world, dist, device)
    # ---- outer proof (rank 0), digest broadcast through the same collective ----
    local = {}
    if rank == 0:
        pis = np.concatenate([digests[0], digests[0]])
        local[0] = proof_digest(prove_fn("outer", 0, 0, pis))
        done += 1
    root = all_gather_digests(local, 1, rank, world, dist, device)[0]
    return root, {"proofs_by_this_rank": done}


class GpuTreeProver:
    """prove_fn backed by nlx_prove: one map circuit, one reduce circuit per level, one outer circuit,
    all resident on this rank's GPU; map witnesses differ per job (seeded), reduce / outer witnesses are
    re-targeted to the children's digests."""

    def __init__(self, nlx, ctx, plan, map_log_n, reduce_log_n, gate_mix=None, torch=None):
        self.nlx, self.ctx, self.plan = nlx, ctx, plan
        mix = gate_mix or dict(pct_poseidon=30, pct_arithmetic=30, pct_base_sum=5, pct_constant=5)
        self.map_syn = nlx.SyntheticCircuit(map_log_n, seed=7001, num_public_inputs=8, **mix)
        self.map_cd = nlx.CircuitData.from_synthetic(ctx, self.map_syn)
        self.red_syn, self.red_cd = [], []
        for lvl in range(len(plan.levels)):
            s = nlx.SyntheticCircuit(reduce_log_n, seed=7100 + lvl, num_public_inputs=8, **mix)
            self.red_syn.append(s)
            self.red_cd.append(nlx.CircuitData.from_synthetic(ctx, s))
        self.out_syn = nlx.SyntheticCircuit(reduce_log_n, seed=7200, num_public_inputs=8, **mix)
        self.out_cd = nlx.CircuitData.from_synthetic(ctx, self.out_syn)

    def __call__(self, kind, level, index, public_inputs):
        if kind == "map":
            syn, cd = self.map_syn, self.map_cd
            pis = np.array([(index * 0x9E3779B97F4A7C15 + k) % P for k in range(8)], dtype=np.uint64)
        elif kind == "reduce":
            syn, cd, pis = self.red_syn[level], self.red_cd[level], public_inputs
        else:
            syn, cd, pis = self.out_syn, self.out_cd, public_inputs
        syn.set_public_inputs(pis)
        return cd.prove(syn.wires, syn.public_inputs)


def bench_verify128(args, nlx, torch, rank, world, local, dist):
    """bench.py --workload verify128: whole VerifyCircuit-128x4-shaped job per step, strong scaling."""
    import time
    plan = TreePlan(32)
    ctx = nlx.Context(local)
    prover = GpuTreeProver(nlx, ctx, plan, args.map_log_n, args.reduce_log_n, torch=torch)
    device = torch.device("cuda", local)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
    root = None
    for _ in range(args.warmup):
        root, _ = run_tree(plan, prover, rank, world, dist, device)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        root, stats = run_tree(plan, prover, rank, world, dist, device)
    sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if rank != 0:
        return None
    return {
        "metric": "Sync/Verify proofs/sec at 1/2/4/8 MI355X + achieved HBM GB/s vs roofline",
        "value": args.steps / dt, "unit": "proofs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "u64 (Goldilocks field, integer)", "data": "synthetic",
        "config": {"workload": "VerifyCircuit 128x4-shaped map-reduce job: 32 map proofs (2^%d rows) + 31 reduce "
                               "proofs + 1 outer proof (2^%d rows), sharded round-robin, one RCCL all-gather of "
                               "digests per level" % (args.map_log_n, args.reduce_log_n),
                   "jobs": plan.n_jobs, "root_digest": [int(x) for x in root], "parallelism": "mapreduce x%d" % world},
        "roofline": None, "cpu_baseline": None,
    }
