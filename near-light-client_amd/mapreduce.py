"""Multi-GPU dispatch of the VerifyCircuit map-reduce proof tree.

Mirrors plonky2x `frontend::mapreduce::generator::MapReduceDynamicGenerator` +
`backend::prover::LocalProver::batch_prove` as used by nearx (`nearx/src/verify.rs:69-90`,
registration at `:112-122`): N/B independent MAP proofs, then a binary tree of REDUCE proofs, then the
outer proof.  The reference runs them one after another on CPU ("No parallelisation", README.md:123);
here the jobs of every level are dealt round-robin over the ranks (one process per GPU) and the only
exchange is ONE all-gather per level (RCCL over xGMI on GPUs, gloo in the CPU tests) of what a parent job
consumes from its children: each child's **blob = public output ‖ serialized proof**.

* A map job proves VERIFY_BATCH ids and outputs a `[ProofVerificationResult; VERIFY_AMT]` array (its own results,
  then defaults - nearx/src/verify.rs:69-86), N x 33 bytes.
* A reduce job takes its two children's proofs and outputs, and outputs their merge (`MergeProofHint`,
  verify.rs:151-183; `succinct_io.merge_verify_outputs`).  In the reference the reduce circuit verifies the two child
  proofs in-circuit; the synthetic reduce circuits here carry the children as public inputs instead: the 2 x 4 field
  elements of `blob_digest(child)` = SHA-256(output ‖ proof), so a reduce proof is bound to exactly those children.
* The outer proof binds the root blob the same way and its output is the job's output (verify.rs:94-98).

Results (root digest, output bytes) do not depend on the number of ranks or on how many jobs are in flight per GPU.
Job counts need not be powers of two (the reference asserts they are, nearx/src/main.rs:19; here an unpaired node of a
level is carried to the next level unchanged).  The prover behind a job is injected (`prove_fn`) so the sharding logic
is testable without a GPU.
"""
import hashlib

import numpy as np

from . import succinct_io

P = 0xFFFFFFFF00000001


def blob_digest(output, proof):
    """4 field elements identifying a child: SHA-256(public output ‖ proof bytes), four little-endian words mod p"""
    h = hashlib.sha256(bytes(output) + bytes(proof)).digest()
    return np.array([int.from_bytes(h[8 * i: 8 * i + 8], "little") % P for i in range(4)], dtype=np.uint64)


class Blob:
    """what travels from a job to its parent: the job's public output and its serialized proof"""
    __slots__ = ("output", "proof")

    def __init__(self, output, proof):
        self.output, self.proof = bytes(output), bytes(proof)

    def digest(self):
        return blob_digest(self.output, self.proof)

    def pack(self):
        return len(self.output).to_bytes(8, "little") + len(self.proof).to_bytes(8, "little") + self.output + self.proof

    @classmethod
    def unpack(cls, raw):
        lo, lp = int.from_bytes(raw[:8], "little"), int.from_bytes(raw[8:16], "little")
        if 16 + lo + lp > len(raw):
            raise ValueError("truncated blob")
        return cls(raw[16:16 + lo], raw[16 + lo:16 + lo + lp])


class TreePlan:
    """Static job list for `n_map` map proofs reduced pairwise to one.  levels[l] = reduce jobs at reduce level l; when
    a level has an odd number of nodes the last one moves up unchanged (carried[l] is True)."""

    def __init__(self, n_map):
        if n_map < 1:
            raise ValueError("at least one map job")
        self.n_map = n_map
        self.levels, self.carried = [], []
        k = n_map
        while k > 1:
            self.levels.append(k // 2)
            self.carried.append(bool(k & 1))
            k = k // 2 + (k & 1)

    @property
    def n_jobs(self):
        return self.n_map + sum(self.levels) + 1  # + outer proof


def owner(job_index, world):
    return job_index % world


def all_gather_blobs(local, n_jobs, rank, world, dist, device=None):
    """local: dict job_index -> Blob.  Returns the list of all n_jobs blobs on every rank.  One collective for the
    lengths and one for the payload (padded to the longest blob of the level): O(100 KB) per job, latency-bound."""
    if dist is None or world == 1:
        return [local[j] for j in range(n_jobs)]
    import torch
    if device is None:   # RCCL moves device tensors only: this rank's own GPU; gloo (CPU tests, rehearsal) host tensors
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    per = (n_jobs + world - 1) // world
    packed = {j: b.pack() for j, b in local.items()}
    lens = torch.zeros(per, dtype=torch.int64, device=device)
    for j, raw in packed.items():
        lens[j // world] = len(raw)
    all_lens = [torch.zeros_like(lens) for _ in range(world)]
    dist.all_gather(all_lens, lens)
    width = max(int(t.max().item()) for t in all_lens)
    send = torch.zeros((per, width), dtype=torch.uint8)
    for j, raw in packed.items():
        send[j // world, :len(raw)] = torch.frombuffer(bytearray(raw), dtype=torch.uint8)
    send = send.to(device)
    recv = [torch.zeros_like(send) for _ in range(world)]
    dist.all_gather(recv, send)
    out = []
    for j in range(n_jobs):
        r, slot = j % world, j // world
        ln = int(all_lens[r][slot].item())
        out.append(Blob.unpack(recv[r][slot, :ln].cpu().numpy().tobytes()))
    return out


def default_request(n_map, batch=4):
    """A synthetic Verify request of n_map * batch ids: (trusted header hash, [id hash]) - the shape of
    nearx/src/verify.rs:47-55 without the account strings, which do not reach the proof tree."""
    header = hashlib.sha256(b"nlx verify request").digest()
    ids = [hashlib.sha256(b"nlx id %d" % i).digest() for i in range(n_map * batch)]
    return header, ids, batch


def reference_request():
    """The Verify request of the reference's own platform record (fixtures/verify_proof.json, kept as
    tests/golden/near/succinct_requests.json): the trusted header hash and the 128 transaction / receipt ids, VERIFY_BATCH = 4
    (nearx/src/config.rs:36-37) - BASELINE.json configs[3]'s input.  None if the fixture is not there."""
    import json
    import os
    from . import nearx_io
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "near", "succinct_requests.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        raw = bytes.fromhex(json.load(f)["verify"]["input"][2:])
    header, ids = nearx_io.decode_verify_input(raw)
    return header, [h for _, h, _ in ids], 4


def run_tree(plan, prove_fn, rank=0, world=1, dist=None, device=None, request=None):
    """Executes the whole tree.  prove_fn(kind, level, index, public_inputs) -> proof bytes, kind "map" | "reduce" |
    "outer", public_inputs = 8 field elements: for a map job a digest of the request slice it proves, for a reduce job
    its children's blob digests, for the outer job the root blob's digest twice.
    Returns (root_digest, stats) on every rank; stats carries the job's output bytes (VERIFY_AMT x 33) and the proofs
    this rank produced."""
    import time
    header, ids, batch = request if request is not None else default_request(plan.n_map)
    n_amt = len(ids)
    if n_amt != plan.n_map * batch:
        raise ValueError("the request must hold n_map * batch ids")
    done = 0
    level_ms, bytes_gathered = [], 0

    def run_level(jobs, outputs):
        """jobs: list of (kind, level, index, public_inputs) owned by this rank; outputs[i]: job i's public output.
        A prover may offer prove_many() to keep several independent jobs of a level in flight."""
        if hasattr(prove_fn, "prove_many"):
            proofs = prove_fn.prove_many(jobs)
        else:
            proofs = [prove_fn(*job) for job in jobs]
        return {job[2]: Blob(out, pr) for job, out, pr in zip(jobs, outputs, proofs)}

    # ---- map level: job j proves ids[j * batch : (j + 1) * batch] ----
    t0 = time.perf_counter()
    jobs, outs = [], []
    for j in range(plan.n_map):
        if owner(j, world) != rank:
            continue
        mine = ids[j * batch:(j + 1) * batch]
        pis = blob_digest(header + b"".join(mine), j.to_bytes(8, "little"))
        jobs.append(("map", 0, j, np.concatenate([pis, pis])))
        results = [(i, True) for i in mine] + succinct_io.default_verify_output(n_amt - len(mine))
        outs.append(succinct_io.encode_verify_output(results))
    local = run_level(jobs, outs)
    done += len(jobs)
    nodes = all_gather_blobs(local, plan.n_map, rank, world, dist, device)
    bytes_gathered += sum(len(b.proof) + len(b.output) for b in nodes)
    level_ms.append(("map", plan.n_map, (time.perf_counter() - t0) * 1e3))
    # ---- reduce levels ----
    for lvl, n_jobs in enumerate(plan.levels):
        t0 = time.perf_counter()
        jobs, outs = [], []
        for j in range(n_jobs):
            if owner(j, world) != rank:
                continue
            left, right = nodes[2 * j], nodes[2 * j + 1]
            jobs.append(("reduce", lvl, j, np.concatenate([left.digest(), right.digest()])))
            merged = succinct_io.merge_verify_outputs(succinct_io.decode_verify_output(left.output),
                                                      succinct_io.decode_verify_output(right.output))
            outs.append(succinct_io.encode_verify_output(merged))
        local = run_level(jobs, outs)
        done += len(jobs)
        nxt = all_gather_blobs(local, n_jobs, rank, world, dist, device)
        bytes_gathered += sum(len(b.proof) + len(b.output) for b in nxt)
        if plan.carried[lvl]:
            nxt.append(nodes[-1])   # the unpaired node moves up unchanged
        nodes = nxt
        level_ms.append(("reduce%d" % lvl, n_jobs, (time.perf_counter() - t0) * 1e3))
    # ---- outer proof (rank 0), broadcast through the same collective ----
    t0 = time.perf_counter()
    local = {}
    root_child = nodes[0]
    if rank == 0:
        d = root_child.digest()
        proof = prove_fn("outer", 0, 0, np.concatenate([d, d]))
        local[0] = Blob(root_child.output, proof)
        done += 1
    if dist is None or world == 1:
        outer = local[0]
    else:
        outer = all_gather_blobs(local, 1, rank, world, dist, device)[0]
    level_ms.append(("outer", 1, (time.perf_counter() - t0) * 1e3))
    return outer.digest(), {"proofs_by_this_rank": done, "level_ms": level_ms, "output": outer.output,
                            "outer_proof": outer.proof, "bytes_gathered": bytes_gathered}


class GpuTreeProver:
    """prove_fn backed by nlx_prove: one map circuit, one reduce circuit per level, one outer circuit,
    all resident on this rank's GPU.  `workers` independent contexts (stream + host thread each) keep
    several jobs of a level in flight.  Witness tables live in HBM (an nlx_buf of the worker's context); a job only
    re-targets the PublicInputGate row (4 words) to its public inputs, through nlx_buf_upload - i.e. on the context's own
    stream, ordered before the proof that reads it (round 1 patched a torch tensor on torch's current stream of whatever
    device the calling thread had current, which nothing ordered against the context's stream)."""

    def __init__(self, nlx, ctx, plan, map_log_n, reduce_log_n, gate_mix=None, torch=None, workers=1):
        import queue
        self.nlx, self.plan = nlx, plan
        mix = gate_mix or dict(pct_poseidon=30, pct_arithmetic=30, pct_base_sum=5, pct_constant=5)
        self.workers = []
        self.free = queue.Queue()
        for w in range(max(1, workers)):
            c = ctx if w == 0 else nlx.Context(ctx.device)
            wk = {"ctx": c, "circ": {}}
            specs = [("map", 0, map_log_n, 7001)] + [("reduce", lvl, reduce_log_n, 7100 + lvl)
                                                     for lvl in range(len(plan.levels))] + [("outer", 0, reduce_log_n, 7200)]
            for kind, lvl, log_n, seed in specs:
                syn = nlx.SyntheticCircuit(log_n, seed=seed, num_public_inputs=8, **mix)
                cd = nlx.CircuitData.from_synthetic(c, syn)
                dev = nlx.DeviceBuffer.from_array(c, syn.wires)
                wk["circ"][(kind, lvl)] = (syn, cd, dev)
            self.workers.append(wk)
            self.free.put(wk)

    def _prove(self, wk, kind, level, index, public_inputs):
        syn, cd, dev = wk["circ"][(kind, level if kind == "reduce" else 0)]
        syn.set_public_inputs(public_inputs)
        n = 1 << syn.log_n
        for col in range(4):   # the PublicInputGate row: wires 0..3 of row 0 hold the public-input hash
            dev.upload(syn.wires[col, 0:1], offset=col * n * 8)
        return cd.prove(dev.ptr, syn.public_inputs)

    def __call__(self, kind, level, index, public_inputs):
        wk = self.free.get()
        try:
            return self._prove(wk, kind, level, index, public_inputs)
        finally:
            self.free.put(wk)

    def prove_many(self, jobs):
        if len(self.workers) == 1 or len(jobs) <= 1:
            return [self(*job) for job in jobs]
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=len(self.workers)) as ex:
            return list(ex.map(lambda job: self(*job), jobs))


def bench_verify128(args, nlx, torch, rank, world, local, dist, verify_outer=None, cpu_baseline=None):
    """bench.py --workload verify128 (and the `verify128` record of the default Sync line): whole VerifyCircuit-128x4-shaped
    job per step, strong scaling.  cpu_baseline: a callable returning the line's cpu_baseline object (bench.py times the
    test oracle there - this module never touches it) or None."""
    import time
    plan = TreePlan(32)
    # the request: the reference's own 128-id Verify request when its fixture is present (every rank reads the same file)
    request = reference_request() or default_request(plan.n_map)
    ctx = nlx.Context(local)
    prover = GpuTreeProver(nlx, ctx, plan, args.map_log_n, args.reduce_log_n, torch=torch, workers=args.inflight)
    device = torch.device("cuda", local)
    if dist is not None and dist.get_backend() == "gloo":  # one-GPU rehearsal of the N > 1 path (bench.py)
        device = torch.device("cpu")

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
    # Optionally the curta sub-proofs of the map jobs as well: a map job verifies 4 transaction / receipt inclusion proofs =
    # 1 444 SHA-256 compression blocks, padded to 2^11.  Each rank proves the blocks of ALL the map
    # jobs it owns in ONE SHA-256 STARK before the tree (one 2^16-block proof on one GPU, 2^13 blocks per GPU on eight):
    # a proof per job would leave the chip mostly idle (2^11 blocks is 2^14 LDE rows; measured 18.6 ms per job against 142 ms
    # for all 32 together), and the binding fingerprint covers every block either way.  Blocks: synthetic 64-byte Merkle nodes.
    stark_ms, last_stark_ms, owned = 0.0, 0.0, 0
    sha = None
    if getattr(args, "map_starks", False):
        sa = nlx.sha256_air
        owned = sum(1 for j in range(plan.n_map) if owner(j, world) == rank)
        lb = 11 + max(0, (owned - 1).bit_length())
        sha = sa.Sha256Prover(ctx, lb)
        rng = np.random.default_rng(17 + rank)
        n_msgs = 1 << (lb - 1)
        raw = rng.integers(0, 256, (n_msgs, 64), dtype=np.uint8)
        pad = np.zeros((n_msgs, 64), dtype=np.uint8)
        pad[:, 0], pad[:, 62] = 0x80, 0x02
        sha_blocks = np.concatenate([raw, pad], axis=1).reshape(2 * n_msgs, 64).view(">u4").astype(np.uint32)
        sha_first = np.tile(np.array([1, 0], dtype=np.uint8), n_msgs)

    def map_starks():
        t1 = time.perf_counter()
        _, digest = sha.generate_trace(sha_blocks, sha_first)
        sha.prove_trace(digest)
        return (time.perf_counter() - t1) * 1e3
    root = None
    for _ in range(args.warmup):
        if sha is not None:
            map_starks()
        root, _ = run_tree(plan, prover, rank, world, dist, device, request)
    sync()
    for wk in prover.workers:
        wk["ctx"].kernel_timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if sha is not None:
            last_stark_ms = map_starks()
            stark_ms += last_stark_ms
        root, stats = run_tree(plan, prover, rank, world, dist, device, request)
    sync()
    dt = time.perf_counter() - t0
    # the dominant kernel of the job = Poseidon leaf hashing of the map proofs' LDE tables (8cL + 32L bytes per launch,
    # SURVEY.md §8d), timed by HIP events on every worker context's own stream inside the timed region
    calls, ms, alg = 0, 0.0, 0.0
    for wk in prover.workers:
        n_, ms_, b_ = wk["ctx"].kernel_stats("hash_lde_leaves")
        calls, ms, alg = calls + n_, ms + ms_, alg + b_
        wk["ctx"].kernel_timing(False)
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if rank != 0:
        return None
    # what the job returned is checked before it is reported: the output lists every requested id as verified, in order
    # (bench.py, which may use the test oracle, additionally runs the oracle verifier on the outer proof)
    header, ids, _ = request
    output_ok = succinct_io.decode_verify_output(stats["output"]) == [(i, True) for i in ids]
    outer_ok = None
    if verify_outer is not None:
        outer_ok = bool(verify_outer(prover.workers[0]["circ"][("outer", 0)][0], stats["outer_proof"]))
    return {
        "metric": "Sync/Verify proofs/sec at 1/2/4/8 MI355X + achieved HBM GB/s vs roofline",
        "value": args.steps / dt, "unit": "proofs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "u64 (Goldilocks field, integer)", "data": "synthetic",
        "config": {"workload": "VerifyCircuit 128x4-shaped map-reduce job: 32 map proofs (2^%d rows) + 31 reduce "
                               "proofs + 1 outer proof (2^%d rows), sharded round-robin, one RCCL all-gather per level of the children's "
                               "(public output || serialized proof) blobs" % (args.map_log_n, args.reduce_log_n) +
                               ("; plus, per rank, one SHA-256 STARK of the 2^11 blocks of every map job it owns" if sha is not None else ""),
                   "map_starks_ms_per_step_rank0": round(stark_ms / args.steps, 2) if sha is not None else None,
                   "jobs": plan.n_jobs, "proofs_in_flight_per_gpu": args.inflight,
                   "map_starks": sha is not None,
                   "level_ms_last_step": ([["map_starks_sha256 (rank 0's %d jobs, 2^%d blocks, one STARK)" % (owned, int(sha.log_blocks)), owned, round(last_stark_ms, 3)]] if sha is not None else []) +
                                         [[k, n, round(ms, 3)] for k, n, ms in stats["level_ms"]], "request": "fixtures/verify_proof.json: 128 real ids under header 0x%s" % request[0].hex() if reference_request() else "synthetic ids",
                   "root_digest": [int(x) for x in root], "bytes_gathered_last_step": stats["bytes_gathered"],
                   "output_bytes": len(stats["output"]), "output_lists_every_id_as_verified": output_ok,
                   "oracle_verifier_accepts_outer_proof": outer_ok, "parallelism": "mapreduce x%d" % world},
        "roofline": None if not calls else {
            "bound": "hbm", "achieved": alg / (ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": alg / (ms * 1e-3) / 1e9 / 8000.0,
            "traffic": None, "kernel": "k_hash_lde_leaves", "launches": calls, "avg_launch_ms": ms / calls,
            "alg_bytes_per_launch": alg / calls,
            "note": "rank 0's launches of the last timed steps, every tree level together (map proofs at 2^%d rows dominate); "
                    "integer-VALU bound like the Sync line's (DESIGN.md §4); with %d proofs in flight the event windows of "
                    "concurrent streams overlap" % (args.map_log_n, args.inflight)},
        "cpu_baseline": cpu_baseline() if cpu_baseline is not None else None,
    }
