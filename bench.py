#!/usr/bin/env python3
"""bench.py - Sync proofs/s of the MI355X-native prover (and the secondary workloads of the same library).

Contract (driver): `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line from
rank 0.  For N > 1 the driver launches it under torch.distributed.run (one rank per GPU, RCCL).

Workloads (--workload):
  sync (default, BASELINE.json configs[1]): a "step" is ONE FULL SYNC PROOF - the SHA-256, SHA-512 and Ed25519 STARKs
    of a real mainnet step (the reference's fixtures main_1 -> main_2; one Ed25519 slot per validator, active where the
    block carries its approval; traces generated on the GPU inside the timed region) and the outer plonky2 proof that
    verifies them: a synthetic circuit of SyncCircuit's static shape - standard_recursion_config (135 wires / 80 routed,
    rate 8, cap height 4, 2 challenges, 28 FRI queries, 16 PoW bits), 2^18 rows (derived from the in-circuit cost of
    verifying the three STARK proofs; --log-n), all nineteen gate kinds (GATE_MIXES below) with real copy
    constraints, a satisfying witness resident in HBM, public inputs = the step's 64 real I/O bytes.
    SyncCircuit does not shard: with N GPUs each rank proves its own independent request
    ("replicas only"), scaling = weak, no data-path collective.
  outer: the outer plonky2 proof alone (round 1's default), --log-n rows, --inflight proofs at a time.
  verify128: the VerifyCircuit 128x4 map-reduce job - 32 map proofs, a binary reduce tree (16+8+4+2+1) and one outer
    proof, sharded over the ranks with one all-gather of the children's (public output || proof) blobs per level
    (scaling = strong).
  stark / sha256 / sha512 / ed25519: one STARK proof per step (synthetic AIR; 2^k SHA-256 / SHA-512 blocks; 2^k Ed25519
    slots), traces generated on the GPU.
  ntt24 [--ntt-field bn254]: 16 columns x 2^24-point NTT (BASELINE.json configs[4]'s transform), over Goldilocks or BN254 Fr.
  msm24: one BN254 G1 multi-scalar multiplication of 2^24 points (the recursive wrap's KZG commitment).

roofline: the dominant kernel is the Poseidon leaf hashing of the LDE tables (k_hash_lde_leaves); its algorithmic bytes
  per launch are 8*c*L + 32*L and its average duration is measured live with HIP events on the launch
  stream; roofline_valu is the bound that binds (integer-VALU issue).
cpu_baseline: the CPU oracle (C port of the same algorithm, OpenMP over the host cores) timed on a bounded sample of the
  same step; the same leg proves the sampled inputs on the GPU and compares the bytes (parity_checked).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
# Integer-VALU issue roofline of the Poseidon permutation kernels (profiles/r02_valu_ubench_v1.txt, r03_poseidon_occupancy.txt).
# Measured on MI355X: every 32 x 32-bit multiply form issues at 1.99 ns per wave-instruction per SIMD (and so does every
# VCC-chained add).  Since round 3 the linear layer's 8 640 multiply-adds run on the matrix cores (eight v_mfma_i32_32x32x32_i8
# per layer over the state's byte planes, beside the vector pipe); what the VECTOR pipe cannot do without is 4 multiplies per
# field multiplication (118 S-boxes x 4) and, per layer, the 96 multiply-adds that put the byte planes' 32-bit sums back
# together (12 outputs x 8 planes): 1 888 + 30 x 96 = 4 768 per permutation.  Peak = what the chip would do if every other
# instruction were free: 1 024 SIMDs x 64 lanes / (4 768 x 1.99 ns) = 6.9 G permutations/s (the matrix pipe's own bound, 240 MFMAs
# of 32 cycles per 64 permutations and SIMD, is ~17 G/s).  The kernel issues 12.1 k vector instructions per permutation at the end of
# round 4 (rounds 4 .. 24 as fused blocks of three partial rounds; 14.2 k before that, 15.7 k in round 3; reductions, carry chains,
# byte transposes: profiles/r04_pmc_traffic_outer_2p18_v8.json), nearly all of them
# at ~1.9 ns - only plain 32-bit add / sub / logic / move issue at 1.1 ns (profiles/r04_valu_issue_rates_v1.txt).  The bound
# below is ROUND 3's, unchanged, so that the fractions of rounds 3 and 4 compare.  Round 2's model (all multiply-adds on the vector
# pipe: 10 528 per permutation, 3.13 G/s) is kept in the note so that the fractions of the two rounds can be compared.
VALU_MULS_PER_PERM = 118 * 4 * 4 + 30 * 96
VALU_NS_PER_MUL = 1.99
VALU_PEAK_GPERM = 1024 * 64 / (VALU_MULS_PER_PERM * VALU_NS_PER_MUL)
VALU_PEAK_NOTE = ("1 024 SIMDs x 64 lanes / (4 768 irreducible vector multiply-adds per permutation - 1 888 in the S-boxes, 96 per layer "
                  "recombining the matrix cores' byte-plane sums - x 1.99 ns measured issue cost per wave-instruction per SIMD, "
                  "tools/ubench/poseidon_ubench.hip -> profiles/r03_poseidon_occupancy.txt; the same bound as in round 3); the permutation "
                  "micro-benchmark itself reaches 2.81 Gperm/s at the end of round 4 (2.35 before the fused partial rounds and the carry-free product, 2.09 - 2.15 in round 3; 1.74 with the linear layer on the vector pipe, "
                  "round 2, whose bound was 10 528 multiply-adds = 3.13 Gperm/s: this line's achieved / 3.13 compares with round 2's fractions)")


def stored_traffic(key, alg_bytes):
    """HBM bytes per launch from the committed PMC pass (profiles/r04_pmc_traffic_outer_2p18_v8.json: FETCH_SIZE x 2 + WRITE_SIZE
    per the microarch guide, separate rocprofv3 --pmc runs of the outer proof at the headline's 2^18 rows, tools/pmc_ratio.py):
    measured ratio traffic / algorithmic bytes x this run's algorithmic bytes.  Not measured in this process (counters need their own run) - labelled "stored"."""
    for name in ("r04_pmc_traffic_outer_2p18_v8.json", "r04_pmc_traffic_outer_2p18_v6.json", "r04_pmc_traffic_outer_2p18_v4.json", "r04_pmc_traffic_outer_2p18_v1.json", "r03_pmc_traffic_outer_2p18.json", "r02_pmc_traffic_v5.json", "r02_pmc_traffic_v2.json"):
        prof = os.path.join(ROOT, "profiles", name)
        if os.path.exists(prof):
            try:
                ratio = json.load(open(prof)).get(key)
                if ratio:
                    return ratio * alg_bytes, "stored: profiles/%s (%s = %.3f)" % (name, key, ratio)
            except Exception:
                pass
    return None, None

# Row shares (percent) of the synthetic SyncCircuit-shaped workload; the rest are NoopGate rows.  "nearx" (default)
# contains all nineteen gate kinds the library implements - Poseidon hashing, base-field and extension arithmetic,
# the recursion gates and plonky2x's u32 / comparison gates - because every gate of a circuit is evaluated at every
# LDE point whatever its share of rows; "basic" is the six-gate mix of the round's earlier measurements.
GATE_MIXES = {
    "nearx": dict(pct_poseidon=25, pct_arithmetic=20, pct_base_sum=5, pct_constant=5, pct_extension=10, pct_misc=10,
                  pct_u32=15),
    "basic": dict(pct_poseidon=30, pct_arithmetic=30, pct_base_sum=5, pct_constant=5),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="sync", choices=["sync", "outer", "verify128", "stark", "sha256", "sha512", "ed25519", "ntt24", "msm24", "plonk24"])
    ap.add_argument("--log-blocks", type=int, default=14, help="sha256 workload: 2^k compression blocks per proof")
    ap.add_argument("--log-slots", type=int, default=10, help="ed25519 workload: 2^k signature slots per proof (>= 4)")
    ap.add_argument("--segment-nodes", type=int, default=None, help="sha256 / sha512 / ed25519 workloads: AIR program segment size in arithmetic nodes (0 = one segment)")
    ap.add_argument("--stark-cols", type=int, default=256)
    ap.add_argument("--log-n", type=int, default=None,
                    help="rows of the (outer) plonky2 proof: default 18 for --workload sync (DESIGN.md §6 derives >= 2^17 from the "
                         "in-circuit cost of verifying the three STARK proofs), 16 for the others")
    ap.add_argument("--gate-mix", default="nearx", choices=["nearx", "basic"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--ntt-log-n", type=int, default=24)
    ap.add_argument("--msm-group", default="g1", choices=["g1", "g2"], help="msm24: the group - G1 (KZG commitments) or G2 (Groth16's B query, coordinates in Fq2)")
    ap.add_argument("--ntt-split", action="store_true", help="ntt24 with N > 1 ranks: split every ONE transform over the ranks (contiguous slices, "
                    "log2 N pairwise slice exchanges) instead of dealing whole columns out")
    ap.add_argument("--ntt-order", default="natural", choices=["natural", "dif", "dit"],
                    help="ntt24 --ntt-field bn254: natural order in and out (default), or gnark-crypto's fft.DIF (bit-reversed out) / fft.DIT (bit-reversed in) - no reordering pass")
    ap.add_argument("--ntt-cols", type=int, default=16)
    ap.add_argument("--ntt-field", default="goldilocks", choices=["goldilocks", "bn254"],
                    help="ntt24 workload: goldilocks (the prover's field) or bn254 (the scalar field of the recursive wrap, row f.4)")
    ap.add_argument("--lookup-tables", type=int, default=0, help="outer workload: plonky2 lookup tables in the circuit (LookupGate / LookupTableGate rows, 0 = none)")
    ap.add_argument("--lookup-bits", type=int, default=16, help="outer workload: 2^k (input, output) pairs per table")
    ap.add_argument("--lookups", type=int, default=100000, help="outer workload: lookups into every table")
    ap.add_argument("--stark-variant", default="starky", choices=["starky", "grouped-leaves"],
                    help="starky (default, the headline): whole-row hash_or_noop Merkle leaves and every opening observed by the "
                         "transcript - the reference's STARK protocol; grouped-leaves: round 3's protocol variant for wide, short traces "
                         "(StarkConfig.grouped()), a labelled extra")
    ap.add_argument("--stark-batch-cols", type=int, default=512,
                    help="starky variant: a commitment round of more than this many columns is committed as several PolynomialBatches of "
                         "at most this many columns (plain plonky2 batches: hash_or_noop leaves over the batch's row, a cap and a FRI oracle "
                         "each - nlx_stark_desc.batch_cols); 0 = one batch per round")
    ap.add_argument("--no-extra", action="store_true", help="sync workload: skip the plonky2-only 2^16 figures and the verify128 record")
    ap.add_argument("--inflight", type=int, default=3,
                    help="independent proofs in flight per GPU (one context + stream + host thread each); the K "
                         "timed steps are shared between them")
    ap.add_argument("--host-witness", default="", choices=["", "pageable", "pinned"],
                    help="hand the witness over as a HOST buffer (PCIe inside the timed region); reported separately "
                         "in DESIGN.md, never the headline value")
    ap.add_argument("--map-log-n", type=int, default=18,
                    help="rows of a map proof; SURVEY.md §8d: 18-20 = Sync / map sized (estimate), 12-14 = reduce sized")
    ap.add_argument("--reduce-log-n", type=int, default=13)
    ap.add_argument("--map-starks", action="store_true", help="verify128 workload: also prove the map jobs' SHA-256 work (1 444 "
                    "compression blocks of 4 inclusion proofs -> 2^11 blocks per job), one STARK per rank over the jobs it owns")
    return ap.parse_args()


def dist_setup(n_gpus):
    """one process per GPU; returns (rank, world, local_rank, torch.distributed or None)"""
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # Rehearsal of the N > 1 path on a one-GPU box: NLX_BENCH_REHEARSAL=1 puts every rank on device 0 and
    # uses gloo (RCCL refuses two ranks on one device).  Never set by the driver; the line says so.
    rehearsal = os.environ.get("NLX_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        if rehearsal:
            dist_mod.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist_mod.init_process_group(backend="nccl", rank=rank, world_size=world,
                                        device_id=torch.device("cuda", local))
        dist = dist_mod
    else:
        torch.cuda.set_device(local)
    assert world == n_gpus, "relaunch_under_torchrun() lets no other combination through"
    return rank, world, local, dist


def reduce_max(dist, torch, dt):
    """MAX over ranks of a host scalar (device tensor for RCCL, host tensor in the gloo rehearsal)."""
    if dist is None:
        return dt
    dev = "cpu" if dist.get_backend() == "gloo" else "cuda"
    tt = torch.tensor([dt], dtype=torch.float64, device=dev)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    return float(tt.item())


def barrier(dist, torch):
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()


def cpu_baseline(nlx, log_n, gate_mix):
    """oracle (port) on a bounded sample: one warm-up proof at 2^(log_n - 4) rows (spins up the OpenMP team, faults
    in the allocator's arenas), then one timed proof at 2^(log_n - 1) rows, scaled linearly in rows"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py
    cores = min(len(os.sched_getaffinity(0)), 16)  # a one-GPU box grants 16 host cores
    os.environ["OMP_NUM_THREADS"] = str(cores)
    sample_log_n = max(log_n - 1, 8)
    warm = nlx.SyntheticCircuit(max(sample_log_n - 3, 6), seed=98, **gate_mix)
    wc = oracle_py.Circuit.from_synthetic(warm)
    wc.prove(warm.wires, warm.public_inputs)
    wc.close()
    syn = nlx.SyntheticCircuit(sample_log_n, seed=99, **gate_mix)
    circ = oracle_py.Circuit.from_synthetic(syn)
    t = time.time()
    proof = circ.prove(syn.wires, syn.public_inputs)
    dt = time.time() - t
    ok = circ.verify(proof) == 1
    circ.close()
    scale = 2.0 ** (log_n - sample_log_n)
    return {"value": 1.0 / (dt * scale), "unit": "proofs/s", "cores": cores, "kind": "port",
            "sample": "1 proof at 2^%d rows (1/%d of the workload's rows) in %.2f s after a warm-up proof, scaled linearly "
                      "in rows; oracle verifier accepted: %s" % (sample_log_n, int(scale), dt, ok)}


def run_outer(args, nlx, torch, rank, world, local, dist):
    """--workload outer: the outer plonky2 proof alone (round 1's default job; --log-n sweeps its size)."""
    import numpy as np
    gate_mix = GATE_MIXES[args.gate_mix]
    lk = dict(num_luts=args.lookup_tables, lut_bits=args.lookup_bits, num_lookups=args.lookups) if args.lookup_tables else {}
    syn = nlx.SyntheticCircuit(args.log_n, seed=1000 + rank, num_public_inputs=64, **gate_mix, **lk)
    # public inputs = the real SyncCircuit I/O of the reference's fixture main_2.json (BASELINE.json configs[0]/[1]): 32-byte trusted
    # header hash in, 32-byte new head hash out (nearx/src/sync.rs:37,43), one field element per byte
    from importlib import import_module
    io = import_module("nlx_amd.nearx_io")
    sync_in, sync_out = io.sync_io(io.load_fixture(os.path.join(ROOT, "tests", "golden", "near", "main_2.json")))
    syn.set_public_inputs(io.bytes_to_field_elements(sync_in + sync_out))
    # witness resident in HBM before the timed region (device tensor handed over by pointer)
    wires = torch.from_numpy(syn.wires.view(np.int64)).cuda()
    if args.host_witness == "pageable":
        wires = torch.from_numpy(syn.wires.view(np.int64))
    elif args.host_witness == "pinned":
        wires = torch.from_numpy(syn.wires.view(np.int64)).pin_memory()
    pis = np.ascontiguousarray(syn.public_inputs)
    pis_ptr = pis.ctypes.data
    # SyncCircuit requests are independent: `inflight` of them are proved concurrently, each on its own
    # context / HIP stream / host thread, so one proof's latency-bound phases (Merkle tops, FRI tails,
    # transcript round trips) overlap another proof's throughput-bound kernels.
    n_workers = max(1, min(args.inflight, args.steps))
    ctxs = [nlx.Context(local) for _ in range(n_workers)]
    cds = [nlx.CircuitData.from_synthetic(c, syn) for c in ctxs]
    for cd in cds:
        for _ in range(args.warmup):
            cd.prove_into(wires, pis_ptr)
    for c in ctxs:
        c.kernel_timing(True)
    # the K timed steps go through ONE C-ABI call (nlx_batch_prove): the library's worker threads take
    # jobs in order, one proof at a time per context
    import ctypes
    cap = nlx.lib.dll.nlx_proof_max_bytes(cds[0].handle)
    bufs = [np.zeros(cap, dtype=np.uint8) for _ in range(args.steps)]
    jobs = (nlx.ProveJob * args.steps)()
    for i in range(args.steps):
        jobs[i].wires = wires.data_ptr()
        jobs[i].public_inputs = pis_ptr
        jobs[i].proof_out = bufs[i].ctypes.data
        jobs[i].proof_cap = cap
    handles = (ctypes.c_void_p * n_workers)(*[cd.handle for cd in cds])
    barrier(dist, torch)
    t0 = time.perf_counter()
    rc = nlx.lib.dll.nlx_batch_prove(handles, n_workers, jobs, args.steps)
    barrier(dist, torch)
    dt = time.perf_counter() - t0
    if rc != 0:
        raise RuntimeError("nlx_batch_prove failed with %d" % rc)
    assert all(jobs[i].proof_len == jobs[0].proof_len for i in range(args.steps))
    dt = reduce_max(dist, torch, dt)
    names = ("intt", "lde", "hash_lde_leaves", "merkle_levels", "quotient", "fri_combine") + (("lookup_terms",) if lk else ())
    kstats = {k: [0, 0.0, 0.0] for k in names}
    for c in ctxs:
        for k in names:
            n_, ms_, b_ = c.kernel_stats(k)
            kstats[k][0] += n_
            kstats[k][1] += ms_
            kstats[k][2] += b_
    calls, ms, alg = kstats["hash_lde_leaves"]
    for c in ctxs:
        c.kernel_timing(False)
    # With several proofs in flight a kernel's event-timed duration includes time-slicing with the
    # other streams.  A short single-stream pass (outside the timed region, same circuit) gives the
    # kernel's own duration; both are reported.
    single = None
    if n_workers > 1:
        ctxs[0].kernel_timing(True)
        for _ in range(2):
            cds[0].prove_into(wires, pis_ptr)
        c1, m1, a1 = ctxs[0].kernel_stats("hash_lde_leaves")
        ctxs[0].kernel_timing(False)
        if c1:
            ach1 = (a1 / c1) / (m1 / c1 * 1e-3) / 1e9
            single = {"achieved": ach1, "frac": ach1 / HBM_PEAK_GBS, "avg_launch_ms": m1 / c1, "launches": c1,
                      "note": "2 proofs, one stream, after the timed region"}
    stages = cds[0].stage_times()
    out = None
    if rank == 0:
        achieved = (alg / calls) / (ms / calls * 1e-3) / 1e9 if calls else 0.0
        traffic = None
        prof = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if os.path.exists(prof):
            try:
                t = json.load(open(prof))
                if t.get("log_n") == args.log_n:
                    traffic = t.get("hash_lde_leaves_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "Sync/Verify proofs/sec at 1/2/4/8 MI355X + achieved HBM GB/s vs roofline",
            "value": world * args.steps / dt, "unit": "proofs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u64 (Goldilocks field, integer)", "data": "synthetic",
            "config": {"workload": "SyncCircuit-shaped plonky2 proof (standard_recursion_config, 2^%d rows, "
                                   "135 wires, rate 8, 28 queries, 16 PoW bits, %d gate kinds), replicas only"
                                   % (args.log_n, syn.num_gates),
                       "log_n": args.log_n, "gate_mix_pct": gate_mix,
                       "lookup_tables": ("%d table(s) of 2^%d pairs, %d lookups each (rows %s): set_lookup_wires, the lookup "
                                         "polynomials and the lookup terms of the quotient inside the timed region"
                                         % (args.lookup_tables, args.lookup_bits, args.lookups, syn.lookup_rows.tolist())) if lk else None,
                       "public_inputs": "64 bytes of real Sync I/O (fixtures/main_2.json): new head hash 0x%s" % sync_out.hex(), "proof_bytes": len(cds[0].prove(wires, pis)),
                       "proofs_in_flight_per_gpu": n_workers, "witness": args.host_witness or "resident in HBM", "parallelism": "replicas x%d" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "k_hash_lde_leaves", "launches": calls, "avg_launch_ms": ms / calls if calls else None,
                         "alg_bytes_per_launch": alg / calls if calls else None,
                         "note": "Poseidon leaf hashing is VALU-integer bound (17 permutations per 135-wide row); "
                                 "the HBM fraction is low by construction; with >1 proof in flight the event-timed "
                                 "duration includes time shared with the other stream's kernels"},
            "roofline_single_stream": single,
            "stage_ms_last_proof": {k: round(v, 3) for k, v in stages},
            "kernel_ms_per_proof": {k: round(v[1] / args.steps, 3) for k, v in kstats.items()},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(nlx, args.log_n, gate_mix)
        else:
            out["cpu_baseline"] = None
    for cd in cds:
        cd.close()
    return out


def run_stark(args, nlx, torch, rank, world, local, dist):
    """Secondary workload: one starky-style proof of the synthetic wide AIR
    (--stark-cols columns x 2^--log-n rows, StarkConfig::standard_fast_config) per step, `inflight`
    independent proofs concurrently (one context + host thread each, nlx_stark_batch_prove)."""
    import numpy as np
    S = nlx.stark
    air = S.wide_air(args.stark_cols, seed=7)
    t, pis = S.wide_trace(air, args.log_n, seed=11 + rank)
    st = S.Stark(air, args.log_n)
    d_t = torch.from_numpy(t.view(np.int64)).cuda()
    n_workers = max(1, min(args.inflight, args.steps))
    ctxs = [nlx.Context(local) for _ in range(n_workers)]
    prs = [st.build(c) for c in ctxs]
    pis_ptr = pis.ctypes.data
    for pr in prs:
        for _ in range(args.warmup):
            pr.prove_into(d_t, pis_ptr)
    for c in ctxs:
        c.kernel_timing(True)
    # the K timed proofs go through ONE C-ABI call (nlx_stark_batch_prove), as the plonky2 workload does
    import ctypes
    cap = nlx.lib.dll.nlx_stark_proof_max_bytes(prs[0].handle)
    bufs = [np.zeros(cap, dtype=np.uint8) for _ in range(args.steps)]
    jobs = (nlx.ProveJob * args.steps)()
    for i in range(args.steps):
        jobs[i].wires = d_t.data_ptr()
        jobs[i].public_inputs = pis_ptr
        jobs[i].proof_out = bufs[i].ctypes.data
        jobs[i].proof_cap = cap
    handles = (ctypes.c_void_p * n_workers)(*[pr.handle for pr in prs])
    barrier(dist, torch)
    t0 = time.perf_counter()
    rc = nlx.lib.dll.nlx_stark_batch_prove(handles, n_workers, jobs, args.steps)
    barrier(dist, torch)
    dt = time.perf_counter() - t0
    if rc != 0:
        raise RuntimeError("nlx_stark_batch_prove failed with %d" % rc)
    dt = reduce_max(dist, torch, dt)
    names = ("intt", "lde", "hash_lde_leaves", "merkle_levels", "air_quotient", "fri_combine")
    kstats = {k: [0, 0.0, 0.0] for k in names}
    for c in ctxs:
        for k in names:
            n_, ms_, b_ = c.kernel_stats(k)
            kstats[k][0] += n_
            kstats[k][1] += ms_
            kstats[k][2] += b_
        c.kernel_timing(False)
    stages = prs[0].stage_times()
    out = None
    if rank == 0:
        calls, ms, alg = kstats["hash_lde_leaves"]
        achieved = (alg / calls) / (ms / calls * 1e-3) / 1e9 if calls else 0.0
        proof = prs[0].prove(d_t, pis)
        out = {
            "metric": "STARK proofs/sec (starky-style, secondary workload)", "value": world * args.steps / dt,
            "unit": "proofs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64 (Goldilocks field, integer)", "data": "synthetic",
            "config": {"workload": "starky-style STARK of the synthetic wide AIR: %d columns x 2^%d rows, degree-3 "
                                   "constraints, standard_fast_config (rate 2, 84 queries, 16 PoW bits), replicas only"
                                   % (args.stark_cols, args.log_n),
                       "air_program_words": int(st.desc.n_words), "constraints": air.num_constraints,
                       "proof_bytes": len(proof), "proofs_in_flight_per_gpu": n_workers, "trace": "resident in HBM"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel": "k_hash_lde_leaves",
                         "launches": calls, "avg_launch_ms": ms / calls if calls else None,
                         "alg_bytes_per_launch": alg / calls if calls else None},
            "stage_ms_last_proof": {k: round(v, 3) for k, v in stages},
            "kernel_ms_per_proof": {k: round(v[1] / args.steps, 3) for k, v in kstats.items()},
        }
        if not args.no_cpu_baseline and world == 1:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import oracle_py
            cores = min(len(os.sched_getaffinity(0)), 16)
            os.environ["OMP_NUM_THREADS"] = str(cores)
            s_log = max(args.log_n - 3, 8)
            t2, pis2 = S.wide_trace(air, s_log, seed=5)
            st2 = S.Stark(air, s_log)
            tc = time.time()
            pr2 = oracle_py.stark_prove(st2.desc, t2, pis2)
            dtc = time.time() - tc
            ok = oracle_py.stark_verify(st2.desc, pr2) == 1
            scale = 2.0 ** (args.log_n - s_log)
            out["cpu_baseline"] = {"value": 1.0 / (dtc * scale), "unit": "proofs/s", "cores": cores, "kind": "port",
                                   "sample": "1 proof at 2^%d rows in %.2f s, scaled linearly in rows; oracle verifier "
                                             "accepted: %s; GPU proof accepted: %s"
                                             % (s_log, dtc, ok, oracle_py.stark_verify(st.desc, proof) == 1)}
        else:
            out["cpu_baseline"] = None
    for pr in prs:
        pr.close()
    for c in ctxs:
        c.close()
    return out


def run_sha256(args, nlx, torch, rank, world, local, dist):
    """Secondary workload: one STARK proof of 2^--log-blocks SHA-256 compression blocks
    per step - trace generation on the GPU (nlx_sha256_trace) + nlx_stark_prove, both inside the timed
    region.  Messages are 64-byte Merkle-node preimages (two blocks each), as nearx's inclusion proofs hash
    (nearx/src/merkle.rs:43-50)."""
    import hashlib
    import struct
    import numpy as np
    wide = args.workload == "sha512"
    SA = nlx.sha512_air if wide else nlx.sha256_air
    rng = np.random.default_rng(5 + rank)
    n_blocks = 1 << args.log_blocks
    if args.log_blocks < 2:
        raise SystemExit("--log-blocks must be >= 2 (a block is four trace rows)")
    if wide:
        # one block per message: R || A || approval message = 105 bytes, the SHA-512 of an Ed25519 check
        n_msgs, msg_len = n_blocks, 105
        raw = rng.integers(0, 256, (n_msgs, msg_len), dtype=np.uint8)
        pad = np.zeros((n_msgs, 128 - msg_len), dtype=np.uint8)
        pad[:, 0] = 0x80
        pad[:, -2], pad[:, -1] = (8 * msg_len) >> 8, (8 * msg_len) & 0xFF
        blocks = np.concatenate([raw, pad], axis=1).reshape(n_msgs, 128).view(">u8").astype(np.uint64)
        first = np.ones(n_msgs, dtype=np.uint8)
        want = np.array(struct.unpack(">8Q", hashlib.sha512(raw[-1].tobytes()).digest()), dtype=np.uint64)
    else:
        n_msgs, msg_len = max(1, n_blocks // 2), 64
        raw = rng.integers(0, 256, (n_msgs, 64), dtype=np.uint8)
        # padded blocks of a 64-byte message: the message, then 0x80 .. length 512
        pad = np.zeros((n_msgs, 64), dtype=np.uint8)
        pad[:, 0] = 0x80
        pad[:, 62] = 0x02
        blocks = np.concatenate([raw, pad], axis=1).reshape(n_msgs * 2, 64).view(">u4").astype(np.uint32)
        first = np.tile(np.array([1, 0], dtype=np.uint8), n_msgs)
        want = np.array(struct.unpack(">8I", hashlib.sha256(raw[-1].tobytes()).digest()), dtype=np.uint64)
    pis_of = SA.digest_halves if wide else (lambda d: d)
    make_prover = SA.Sha512Prover if wide else SA.Sha256Prover
    ctx = nlx.Context(local)
    sp = make_prover(ctx, args.log_blocks, nlx.StarkConfig(batch_cols=args.stark_batch_cols), segment_nodes=args.segment_nodes)

    def prove_current(pis):
        """prove the trace generate_trace() left on the device (two rounds: the second is the binding accumulator)"""
        return sp.prove_trace(pis)
    digest = None
    for _ in range(args.warmup):
        trace, digest = sp.generate_trace(blocks, first)
        pis = pis_of(digest)
        prove_current(pis)
    ctx.kernel_timing(True)
    barrier(dist, torch)
    t0 = time.perf_counter()
    t_trace = 0.0
    for _ in range(args.steps):
        t1 = time.perf_counter()
        trace, digest = sp.generate_trace(blocks, first)
        t_trace += time.perf_counter() - t1
        pis = pis_of(digest)
        prove_current(pis)
    barrier(dist, torch)
    dt = time.perf_counter() - t0
    dt = reduce_max(dist, torch, dt)
    names = ("intt", "lde", "hash_lde_leaves", "merkle_levels", "air_quotient", "fri_combine")
    kstats = {k: ctx.kernel_stats(k) for k in names}
    ctx.kernel_timing(False)
    stages = sp.prover.stage_times()
    out = None
    if rank == 0:
        assert np.array_equal(digest, want), "GPU chaining value is not the real digest"
        calls, ms, alg = kstats["hash_lde_leaves"]
        achieved = (alg / calls) / (ms / calls * 1e-3) / 1e9 if calls else 0.0
        proof = sp.prove_trace(pis_of(digest))
        n_rows = SA.ROWS_PER_BLOCK << args.log_blocks
        out = {
            "metric": "%s STARK: compression blocks proved per second (secondary workload)" % ("SHA-512" if wide else "SHA-256"),
            "value": world * args.steps * n_blocks / dt, "unit": "blocks/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64 (Goldilocks field, integer)", "data": "synthetic",
            "config": {"workload": "STARK of 2^%d %s compression blocks (%d rows x %d columns, degree-3 AIR, "
                                   "standard_fast_config: rate 2, 84 queries, 16 PoW bits); trace generated on the GPU "
                                   "inside the timed region; replicas only"
                                   % (args.log_blocks, "SHA-512" if wide else "SHA-256", n_rows, SA.N_COLS),
                       "messages": "%d random %d-byte messages (%s)" % (n_msgs, msg_len, "1 block each, the shape of an Ed25519 "
                                                                         "hash of a NEAR approval" if wide else "2 blocks each"),
                       "air_program_words": int(sp.stark.desc.n_words), "constraints": sp.stark.air.num_constraints,
                       "proof_bytes": len(proof), "trace_gen_ms_per_step": t_trace / args.steps * 1e3,
                       "trace_bytes": int(SA.N_COLS) * n_rows * 8},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel": "k_hash_lde_leaves",
                         "launches": calls, "avg_launch_ms": ms / calls if calls else None,
                         "alg_bytes_per_launch": alg / calls if calls else None},
            "stage_ms_last_proof": {k: round(v, 3) for k, v in stages},
            "kernel_ms_per_proof": {k: round(v[1] / args.steps, 3) for k, v in kstats.items()},
        }
        if not args.no_cpu_baseline and world == 1:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import oracle_py
            cores = min(len(os.sched_getaffinity(0)), 16)
            os.environ["OMP_NUM_THREADS"] = str(cores)
            s_lb = max(min(args.log_blocks - 3, 9), 2)
            sp2 = make_prover(ctx, s_lb)
            b2, f2 = blocks[: 1 << s_lb], first[: 1 << s_lb].copy()
            f2[0] = 1
            tr2, dg2 = sp2.generate_trace(b2, f2)
            host_trace = tr2.cpu().numpy().view(np.uint64)
            tc = time.time()
            pr2 = oracle_py.stark_prove_rounds(sp2.stark.desc, SA.cpu_rounds(b2, f2, host_trace), pis_of(dg2))
            dtc = time.time() - tc
            ok = oracle_py.stark_verify(sp2.stark.desc, pr2) == 1
            sp2.close()
            out["cpu_baseline"] = {"value": (1 << s_lb) / dtc, "unit": "blocks/s", "cores": cores, "kind": "port",
                                   "sample": "oracle STARK prover on 2^%d blocks in %.2f s (trace taken from the GPU "
                                             "generator, not timed); oracle verifier accepted: %s; oracle verifier on "
                                             "the GPU proof of the full workload: %s"
                                             % (s_lb, dtc, ok, oracle_py.stark_verify(sp.stark.desc, proof) == 1)}
        else:
            out["cpu_baseline"] = None
    sp.close()
    ctx.close()
    return out


def run_ed25519(args, nlx, torch, rank, world, local, dist):
    """Secondary workload: one STARK proof of 2^--log-slots Ed25519 verifications per step - trace
    generation (nlx_ed25519_trace), multiplicities and lookup columns (nlx_logup_*) and the two-round STARK, all on the
    GPU inside the timed region.  Statements: synthetic true (A, R, S, h) tuples (64 distinct, tiled)."""
    import numpy as np
    E = nlx.ed25519_air
    n_slots = 1 << args.log_slots
    distinct = E.synthetic_slots(64, seed=9 + rank)
    words = np.tile(E.slots_to_words(distinct), (n_slots // 64, 1))
    ctx = nlx.Context(local)
    pr = E.Ed25519Prover(ctx, args.log_slots, nlx.StarkConfig(batch_cols=args.stark_batch_cols), segment_nodes=args.segment_nodes)
    for _ in range(args.warmup):
        pr.prove(words)
    ctx.kernel_timing(True)
    barrier(dist, torch)
    t0 = time.perf_counter()
    t_trace = 0.0
    for _ in range(args.steps):
        t1 = time.perf_counter()
        pr.generate_trace(words)
        t_trace += time.perf_counter() - t1
        proof = pr.prover.prove_rounds(lambda rnd, known: pr._t0 if rnd == 0 else pr.round1(known), [])
    barrier(dist, torch)
    dt = time.perf_counter() - t0
    dt = reduce_max(dist, torch, dt)
    names = ("intt", "lde", "hash_lde_leaves", "merkle_levels", "air_quotient", "fri_combine")
    kstats = {k: ctx.kernel_stats(k) for k in names}
    ctx.kernel_timing(False)
    stages = pr.prover.stage_times()
    out = None
    if rank == 0:
        calls, ms, alg = kstats["hash_lde_leaves"]
        achieved = (alg / calls) / (ms / calls * 1e-3) / 1e9 if calls else 0.0
        n_rows = n_slots * E.ROWS
        out = {
            "metric": "Ed25519 STARK: signature verifications proved per second (secondary workload)",
            "value": world * args.steps * n_slots / dt, "unit": "signatures/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64 (Goldilocks field, integer)", "data": "synthetic",
            "config": {"workload": "two-round STARK of 2^%d Ed25519 verifications (%d rows x %d + %d columns, degree-3 AIR, 16 "
                                   "multiplication units mod 2^255-19 per row, %d + %d range-check lookups per row (2^16 / 2^9 tables); standard_fast_config: "
                                   "rate 2, 84 queries, 16 PoW bits); trace, multiplicities and lookup columns generated on the "
                                   "GPU inside the timed region; replicas only"
                                   % (args.log_slots, n_rows, E.N_COLS0, E.N_COLS1, len(E.LOOKUPS), len(E.LOOKUPS9)),
                       "air_program_words": int(pr.stark.desc.n_words), "constraints": pr.es.air.num_constraints,
                       "proof_bytes": len(proof), "trace_gen_ms_per_step": t_trace / args.steps * 1e3,
                       "trace_bytes": int(E.N_COLS0 + E.N_COLS1) * n_rows * 8},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel": "k_hash_lde_leaves",
                         "launches": calls, "avg_launch_ms": ms / calls if calls else None,
                         "alg_bytes_per_launch": alg / calls if calls else None},
            "stage_ms_last_proof": {k: round(v, 3) for k, v in stages},
            "kernel_ms_per_proof": {k: round(v[1] / args.steps, 3) for k, v in kstats.items()},
        }
        if not args.no_cpu_baseline and world == 1:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import oracle_py
            cores = min(len(os.sched_getaffinity(0)), 16)
            os.environ["OMP_NUM_THREADS"] = str(cores)
            pr2 = pr if args.log_slots == 8 else E.Ed25519Prover(ctx, 8)
            host = pr2.generate_trace(np.tile(words, (4, 1))[:256]).cpu().numpy().view(np.uint64)
            tc = time.time()
            def cpu_round1(known):
                acc, total = E.binding_columns(host, known[2:4])
                return np.concatenate([oracle_py.logup_round(host, E.LOOKUPS, 16, host[E.MULT], known[:2]),
                                       oracle_py.logup_round(host, E.LOOKUPS9, 9, host[E.MULT9], known[:2]), acc], axis=0), list(total)
            p2 = oracle_py.stark_prove_rounds(pr2.stark.desc, lambda rnd, known: host if rnd == 0 else cpu_round1(known), [])
            dtc = time.time() - tc
            out["cpu_baseline"] = {"value": 256 / dtc, "unit": "signatures/s", "cores": cores, "kind": "port",
                                   "sample": "oracle two-round STARK prover (incl. its lookup columns) on 2^8 slots in %.1f s (trace "
                                             "taken from the GPU generator, not timed); oracle verifier accepted it: %s; oracle "
                                             "verifier on the GPU proof of the full workload: %s"
                                             % (dtc, oracle_py.stark_verify(pr2.stark.desc, p2) == 1,
                                                oracle_py.stark_verify(pr.stark.desc, proof) == 1)}
            if pr2 is not pr:
                pr2.close()
        else:
            out["cpu_baseline"] = None
    pr.close()
    ctx.close()
    return out


def sync_step_setup(args, nlx, torch, rank, local):
    """Everything one real Sync step needs, resident on GPU `local`: the step main_1 -> main_2 of the reference's mainnet
    fixtures (BASELINE.json configs[0]/[1] name fixtures/main_2.json; crates/protocol/src/lib.rs:364-405 walks the same
    chain).  What nearx proves for it (nearx/src/builder.rs:265-308 `sync`): header / next_bps hashing = curta_sha256
    (builder.rs:220,316; variables.rs:71-72), one SHA-512 + one Ed25519 verification per signed approval =
    curta_eddsa_verify_sigs_conditional (builder.rs:116-164), and the outer plonky2 circuit that verifies those STARKs."""
    import json
    import numpy as np
    NP, SA, SB, E = nlx.near_protocol, nlx.sha256_air, nlx.sha512_air, nlx.ed25519_air
    near = os.path.join(ROOT, "tests", "golden", "near")
    with open(os.path.join(near, "main_1.json")) as f:
        bps = json.load(f)["body"]["next_bps"]
    with open(os.path.join(near, "main_2.json")) as f:
        nxt = json.load(f)["body"]
    sha_msgs = NP.sync_sha256_messages(nxt)
    # one Ed25519 slot per validator of the epoch (validate_signatures<LEN>): active where the block carries its approval
    stmt = NP.approval_statement(bps, nxt)
    sig_msgs, slots = stmt["sig_msgs"], stmt["slots"]
    n_sigs = len(sig_msgs)
    st = {"n_sigs": n_sigs, "n_validators": len(slots), "sha_msgs": sha_msgs, "sig_msgs": sig_msgs, "slots": slots, "nxt": nxt,
          "statement": stmt}
    st["lb256"] = max(2, (sum(len(SA.pad_message(m)) for m in sha_msgs) - 1).bit_length())
    st["lb512"] = max(2, (n_sigs - 1).bit_length())
    st["log_slots"] = max(4, (len(slots) - 1).bit_length())   # below 2^8 slots the range table is spread over several columns
    st["bound_slots"] = slots + [E.inactive_slot()] * ((1 << st["log_slots"]) - len(slots))
    st["slot_words"] = E.slots_to_words(st["bound_slots"])
    # the four proofs get a context (HIP stream + scratch) each; the three STARK transcripts open with the step's tag
    # (a hash of the step's 64 public I/O bytes - the outer proof's public inputs)
    io = nlx.nearx_io
    sync_in, sync_out = io.sync_io(io.load_fixture(os.path.join(near, "main_2.json")))
    st["step_tag"] = nlx.stark.step_tag(sync_in + sync_out)
    ctxs = [nlx.Context(local) for _ in range(4)]
    st["ctxs"] = ctxs
    variant = getattr(args, "stark_variant", "starky")
    batch_cols = 0 if variant == "grouped-leaves" else int(getattr(args, "stark_batch_cols", 512))
    mk_cfg = nlx.StarkConfig.grouped if variant == "grouped-leaves" else (lambda: nlx.StarkConfig(batch_cols=batch_cols))
    st["stark_variant"], st["stark_batch_cols"] = variant, batch_cols
    st["p256"] = SA.Sha256Prover(ctxs[0], st["lb256"], mk_cfg(), step_tag=st["step_tag"])
    st["p512"] = SB.Sha512Prover(ctxs[1], st["lb512"], mk_cfg(), step_tag=st["step_tag"])
    st["ped"] = E.Ed25519Prover(ctxs[2], st["log_slots"], mk_cfg(), step_tag=st["step_tag"])
    syn = nlx.SyntheticCircuit(args.log_n, seed=1000 + rank, num_public_inputs=64, **GATE_MIXES[args.gate_mix])
    syn.set_public_inputs(io.bytes_to_field_elements(sync_in + sync_out))
    st["syn"], st["sync_out"] = syn, sync_out
    st["cd"] = nlx.CircuitData.from_synthetic(ctxs[3], syn)
    st["wires"] = torch.from_numpy(syn.wires.view(np.int64)).to("cuda:%d" % local)   # witness resident in HBM
    st["pis"] = np.ascontiguousarray(syn.public_inputs)
    return st


def stark_verifier_rows(st):
    """Lower bound on the rows of the outer circuit, from what it must do: verify the three STARK proofs.  One
    PoseidonGate row per permutation (plonky2 hashes in-circuit with one gate per permutation): per FRI query and per
    committed oracle, ceil(cols / 8) permutations for the opened leaf and one per Merkle sibling; per FRI reduction round a
    2^arity-point leaf of extension values (2 * 2^arity / 8 permutations) and its path; plus the transcript (every opened
    value is observed: 2 * (2 * cols + quotient) elements / 8).  Arithmetic rows (FRI folding, the AIR's constraints at zeta)
    come on top and are not counted, so this is a floor.  DESIGN.md §6 works the numbers."""
    rows = {}
    for name, pr in (("sha256", st["p256"]), ("sha512", st["p512"]), ("ed25519", st["ped"])):
        d = pr.stark.desc
        log_l = d.degree_bits + d.rate_bits
        n_rounds = d.n_rounds if d.n_rounds else 1
        oracles = [d.round_cols[r] for r in range(n_rounds)] if d.n_rounds else [d.n_cols]
        if d.batch_cols:   # a round of more than batch_cols columns is several oracles: a leaf and a Merkle path each
            B = d.batch_cols
            oracles = [w for c in oracles for w in ([B] * (c // B) + ([c % B] if c % B else []) if c > B else [c])]
        oracles.append(d.num_challenges * d.quotient_degree_factor)
        G = d.leaf_group_cols

        def leaf_perms(c):
            """permutations to hash one opened row: whole-row hash_or_noop, or (grouped leaves) every run of G columns and
            then the runs' digests"""
            if G and c > G:
                k = -(-c // G)
                return (k - 1) * ((G + 7) // 8) + (c - (k - 1) * G + 7) // 8 + (4 * k + 7) // 8
            return (c + 7) // 8 if c > 4 else 0
        per_query = sum(leaf_perms(c) for c in oracles) + len(oracles) * (log_l - d.cap_height)
        bits, fri_rounds = d.degree_bits, 0
        while bits > d.fri_final_poly_bits and bits + d.rate_bits >= d.cap_height + d.fri_arity_bits and bits >= d.fri_arity_bits:
            bits -= d.fri_arity_bits
            fri_rounds += 1
        ll = log_l
        for _ in range(fri_rounds):
            ll -= d.fri_arity_bits
            per_query += (2 << d.fri_arity_bits) // 8 + max(ll - d.cap_height, 0)
        n_observed = 2 * (2 * d.n_cols + oracles[-1])
        transcript = (n_observed + 7) // 8
        if d.openings_group:   # the runs' digests, then the digest of those
            k = -(-n_observed // d.openings_group)
            transcript = k * ((d.openings_group + 7) // 8) + (4 * k + 7) // 8
        rows[name] = per_query * d.fri_num_queries + transcript
    rows["total"] = sum(rows.values())
    return rows


def run_sync(args, nlx, torch, rank, world, local, dist):
    """Default workload (BASELINE.json configs[1]): ONE FULL SYNC PROOF per step = the SHA-256 STARK, the SHA-512 STARK
    and the Ed25519 STARK of a real mainnet step (traces generated on the GPU inside the timed region) + the outer plonky2
    proof (2^--log-n rows, default 2^18).  Within a step the outer proof consumes the STARK proofs, so it starts only after
    all three are done (they run concurrently: own context, stream and host thread each); successive Sync requests are
    independent ("replicas only"), so step i+1's STARKs overlap step i's outer proof."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    st = sync_step_setup(args, nlx, torch, rank, local)
    p256, p512, ped, cd, wires, pis = st["p256"], st["p512"], st["ped"], st["cd"], st["wires"], st["pis"]
    ctxs = st["ctxs"]
    outer_buf = cd._buf   # nlx_prove writes the proof here (prove_into returns its length)
    stark_jobs = [lambda: p256.prove(st["sha_msgs"]), lambda: p512.prove(st["sig_msgs"]), lambda: ped.prove(st["slot_words"])]
    pool = ThreadPoolExecutor(max_workers=6)   # three STARK threads, the running outer proof, one queued behind it

    def outer_job():
        return cd.prove_into(wires, pis.ctypes.data)

    def timed(fn, slot, acc):
        t1 = time.perf_counter()
        r = fn()
        acc[slot] += 1e3 * (time.perf_counter() - t1)
        return r

    def run_steps(k, pipelined, acc):
        """k Sync proofs; returns the last step's (sha256, sha512, ed25519) results and outer proof length"""
        res, outer_len, pending = None, 0, []

        def outer_after(prev, acc_):
            if prev is not None:
                prev.result()
            return timed(outer_job, 3, acc_)
        for _ in range(k):
            if pipelined:
                futs = [pool.submit(timed, j, i, acc) for i, j in enumerate(stark_jobs)]
                res = [f.result() for f in futs]            # this request's STARKs (earlier requests' outer proofs run meanwhile)
                # its outer proof: queued behind the previous request's (one context proves one at a time), NOT waited for
                # here - the next request's STARKs start at once; at most two outer proofs are ever outstanding
                if len(pending) >= 2:
                    outer_len = pending.pop(0).result()
                pending.append(pool.submit(outer_after, pending[-1] if pending else None, acc))
            else:
                res = [timed(j, i, acc) for i, j in enumerate(stark_jobs)]
                outer_len = timed(outer_job, 3, acc)
        for f in pending:
            outer_len = f.result()
        return res, outer_len
    scratch = [0.0] * 4
    if args.warmup:
        run_steps(args.warmup, True, scratch)
    for c in ctxs:
        c.kernel_timing(True)
    barrier(dist, torch)
    t0 = time.perf_counter()
    (a, b, c_proof), outer_len = run_steps(args.steps, True, [0.0] * 4)
    barrier(dist, torch)
    dt = reduce_max(dist, torch, time.perf_counter() - t0)
    names = ("intt", "lde", "hash_lde_leaves", "merkle_levels", "quotient", "air_quotient", "fri_combine")

    def collect(which=None):
        ks = {k: [0, 0.0, 0.0, 0.0] for k in names}
        for c in (ctxs if which is None else which):
            for k in names:
                n_, ms_, b_ = c.kernel_stats(k)
                ks[k][0] += n_
                ks[k][1] += ms_
                ks[k][2] += b_
                ks[k][3] += c.kernel_units(k)
        return ks
    kstats = collect()
    kouter = collect(ctxs[3:4])   # the outer proof's launches: the dominant kernel's dominant shape (2^21 leaves x 135 / 20 / 16 columns)
    for c in ctxs:
        c.kernel_timing(False)
    out = None
    if rank == 0:
        import hashlib
        import struct
        io = nlx.nearx_io
        outer_proof = outer_buf[:outer_len].tobytes()
        # the public digests are the real ones: next_bp_hash of the header, hashlib's SHA-512 of the last approval
        assert b"".join(struct.pack(">I", int(x)) for x in a[1]) == io.b58decode32(st["nxt"]["inner_lite"]["next_bp_hash"])
        assert [int(x) for x in b[1]] == list(struct.unpack(">8Q", hashlib.sha512(st["sig_msgs"][-1]).digest()))
        # one step with one proof at a time, kernel events on: per-proof times and the kernels' own durations
        for c in ctxs:
            c.kernel_timing(True)
        parts = [0.0] * 4
        t1 = time.perf_counter()
        run_steps(2, False, parts)
        seq_ms = (time.perf_counter() - t1) / 2 * 1e3
        ks1, ks1_outer = collect(), collect(ctxs[3:4])
        for c in ctxs:
            c.kernel_timing(False)
        parts = [p / 2 for p in parts]

        def roofs(ks):
            calls, ms, alg, perms = ks["hash_lde_leaves"]
            if not calls:
                return None, None
            ach = alg / (ms * 1e-3) / 1e9
            gperm = perms / (ms * 1e-3) / 1e9
            hbm = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                   "traffic": None, "kernel": "k_hash_lde_leaves", "launches": calls, "avg_launch_ms": ms / calls,
                   "alg_bytes_per_launch": alg / calls}
            valu = {"bound": "integer VALU issue", "achieved": gperm, "peak": VALU_PEAK_GPERM, "unit": "Gperm/s",
                    "frac": gperm / VALU_PEAK_GPERM, "kernel": "k_hash_lde_leaves", "permutations_per_launch": perms / calls,
                    "peak_model": VALU_PEAK_NOTE}
            return hbm, valu
        hbm, valu = roofs(kouter)          # roofline of the dominant kernel = the outer proof's leaf hashing ...
        hbm_all, valu_all = roofs(kstats)  # ... and averaged over every launch of the four proofs (small latency-bound STARK launches included)
        hbm1, valu1 = roofs(ks1_outer)
        if hbm is not None:
            hbm["traffic"], hbm["traffic_source"] = stored_traffic("hash_lde_leaves_fetch_over_algorithmic", hbm["alg_bytes_per_launch"])
            hbm["note"] = ("Poseidon leaf hashing of the outer proof's three LDE tables (8cL + 32L bytes per launch, SURVEY.md §8d; 135 / 20 / 16 columns x 2^%d rows); " % (args.log_n + 3) +
                           "integer-VALU bound, so the HBM fraction is low by construction - see roofline_valu; with several "
                           "proofs in flight the event-timed duration includes time shared with the other streams' kernels "
                           "(roofline_single_stream: one proof at a time)")
        rows = stark_verifier_rows(st)
        out = {
            "metric": "Sync/Verify proofs/sec at 1/2/4/8 MI355X + achieved HBM GB/s vs roofline",
            "value": world * args.steps / dt, "unit": "proofs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u64 (Goldilocks field, integer)",
            "data": "synthetic outer circuit of SyncCircuit's static shape; the three STARK statements are the real mainnet Sync step main_1 -> main_2 (the reference's fixtures), traces generated on the GPU",
            "config": {"workload": "one full Sync proof = SHA-256 STARK of %d header / next_bps messages (2^%d blocks, 2^%d x %d trace) + "
                                   "SHA-512 STARK of %d approval hashes (2^%d blocks) + Ed25519 STARK of %d validator slots, %d of them active = "
                                   "signed approvals, D mod L reduced in the AIR (2^%d slots, 2^%d rows) + outer plonky2 proof (standard_recursion_config, 2^%d rows, 135 wires, %d gate "
                                   "kinds); the outer proof of a request starts when its three STARKs are done (and the previous request's outer proof, "
                                   "which shares its context); the next requests' STARKs overlap it, at most two outer proofs outstanding; replicas only"
                                   % (len(st["sha_msgs"]), st["lb256"], st["lb256"] + 2, p256.stark.desc.n_cols, st["n_sigs"], st["lb512"],
                                      st["n_validators"], st["n_sigs"], st["log_slots"], ped.stark.desc.degree_bits, args.log_n, st["syn"].num_gates),
                       "log_n_outer": args.log_n, "gate_mix_pct": GATE_MIXES[args.gate_mix],
                       # the STARK protocol of the three sub-proofs: "starky" = the reference's (whole-row hash_or_noop leaves,
                       # every opening observed); 0 / 0 below is the only setting an unpatched starky-style verifier accepts
                       "stark_variant": st["stark_variant"],
                       "leaf_group_cols": max(int(pr.stark.desc.leaf_group_cols) for pr in (p256, p512, ped)),
                       "openings_group": max(int(pr.stark.desc.openings_group) for pr in (p256, p512, ped)),
                       # a commitment round of more columns than this is several PolynomialBatches (each with whole-row
                       # hash_or_noop leaves, its own cap and FRI oracle): plonky2's own batching, more Merkle paths for the verifier
                       "stark_batch_cols": st["stark_batch_cols"],
                       "outer_rows_floor_from_stark_verification": rows,
                       "public_inputs": "64 bytes of real Sync I/O (fixtures/main_2.json): new head hash 0x%s" % st["sync_out"].hex(),
                       "proof_bytes": {"sha256": len(a[0]), "sha512": len(b[0]), "ed25519": len(c_proof), "outer": outer_len},
                       "parallelism": "replicas x%d" % world},
            "roofline": hbm, "roofline_valu": valu,
            "roofline_all_launches": {"hbm_frac": hbm_all["frac"] if hbm_all else None, "valu_frac": valu_all["frac"] if valu_all else None,
                                      "launches": hbm_all["launches"] if hbm_all else None,
                                      "note": "k_hash_lde_leaves averaged over every launch of the four proofs, the STARKs' small latency-bound ones included"},
            "roofline_single_stream": {"hbm_frac": hbm1["frac"] if hbm1 else None, "hbm_achieved": hbm1["achieved"] if hbm1 else None,
                                       "valu_frac": valu1["frac"] if valu1 else None, "valu_achieved": valu1["achieved"] if valu1 else None,
                                       "avg_launch_ms": hbm1["avg_launch_ms"] if hbm1 else None, "launches": hbm1["launches"] if hbm1 else None,
                                       "note": "2 steps, one proof at a time, after the timed region"},
            "ms_one_proof_at_a_time": {"sha256": round(parts[0], 2), "sha512": round(parts[1], 2), "ed25519": round(parts[2], 2),
                                       "outer_plonky2": round(parts[3], 2), "step": round(seq_ms, 2)},
            # additive per-step costs: the one-proof-at-a-time pass after the timed region (two steps, HIP events on each
            # context's own stream).  The events of the TIMED region bracket kernels of four concurrent streams: those windows
            # overlap and must not be read as shares of ms_per_step - kept under their own name.
            "kernel_ms_per_step": {k: round(v[1] / 2, 3) for k, v in ks1.items()},
            "kernel_event_ms_overlapping": {k: round(v[1] / args.steps, 3) for k, v in kstats.items()},
            "outer_stage_ms_last_proof": {k: round(v, 3) for k, v in cd.stage_times()},
        }
        # the plonky2 proof alone at 2^16 rows (round 1's headline shape), same code, one proof at a time and three in flight
        out["plonky2_only"] = {"log_n_%d_ms_single_stream" % args.log_n: round(parts[3], 2)}
        if not args.no_extra:
            out["plonky2_only"].update(outer_only_figures(nlx, ctxs, local, torch, rank, 16, args.gate_mix))
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"], out["parity_checked"] = cpu_baseline_sync(args, nlx, st, a, b, c_proof, seq_ms)
        else:
            out["cpu_baseline"] = None
        out["outer_proof_sha256"] = hashlib.sha256(outer_proof).hexdigest()
    pool.shutdown()
    for pr in (p256, p512, ped):
        pr.close()
    cd.close()
    for c_ in ctxs:
        c_.close()
    # BASELINE.json configs[3] in the same line, so that the driver's N = 1, 2, 4, 8 runs of this command carry the STRONG-scaling
    # figure north_star asks for next to the Sync replicas: the 128 x 4 Verify job (32 map + 31 reduce + 1 outer proofs, sharded
    # round-robin, one RCCL all-gather of the children's blobs per level), 1 warm-up + 2 timed jobs.  Every rank takes part.
    if not args.no_extra:
        import copy
        va = copy.copy(args)
        va.steps, va.warmup, va.map_starks = 2, 1, True   # the map jobs' SHA-256 STARKs are part of the recorded job (builder.rs:344-363, merkle.rs:43-50)
        try:
            v = run_verify128(va, nlx, torch, rank, world, local, dist)
        except Exception as e:   # the Sync measurement stands whatever happens to the extra record (every rank runs the same code,
            v = None             # so a deterministic failure is raised on all of them and no collective is left waiting)
            if out is not None:
                out["verify128"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if out is not None and v is not None:
            out["verify128"] = {"proofs_per_s": v["value"], "ms_per_job": v["ms_per_step"], "n_gpus": world, "scaling": "strong",
                                "level_ms": v["config"]["level_ms_last_step"], "bytes_gathered": v["config"]["bytes_gathered_last_step"],
                                "root_digest": v["config"]["root_digest"], "output_ok": v["config"]["output_lists_every_id_as_verified"],
                                "oracle_verifier_accepts_outer_proof": v["config"]["oracle_verifier_accepts_outer_proof"],
                                "map_starks": v["config"]["map_starks"], "map_starks_ms_per_job": v["config"]["map_starks_ms_per_step_rank0"],
                                "map_log_n": va.map_log_n, "reduce_log_n": va.reduce_log_n, "proofs_in_flight_per_gpu": va.inflight,
                                "roofline": v["roofline"], "cpu_baseline": v["cpu_baseline"],
                                "note": "the VerifyCircuit 128 x 4 map-reduce job of `--workload verify128`, 1 warm-up + 2 timed jobs after "
                                        "the Sync measurement; strong scaling: the same 64 proofs whatever the number of GPUs"}
    return out


def outer_only_figures(nlx, ctxs, local, torch, rank, log_n, gate_mix):
    """ms per plonky2 proof at 2^log_n rows: one at a time on one context, and three in flight (nlx_batch_prove)"""
    import ctypes
    import numpy as np
    syn = nlx.SyntheticCircuit(log_n, seed=1000 + rank, num_public_inputs=64, **GATE_MIXES[gate_mix])
    cds = [nlx.CircuitData.from_synthetic(c, syn) for c in ctxs[:3]]
    wires = torch.from_numpy(syn.wires.view(np.int64)).to("cuda:%d" % local)
    pis = np.ascontiguousarray(syn.public_inputs)
    for cd in cds:
        cd.prove_into(wires, pis.ctypes.data)
    t0 = time.perf_counter()
    for _ in range(4):
        cds[0].prove_into(wires, pis.ctypes.data)
    single = (time.perf_counter() - t0) / 4 * 1e3
    k = 12
    cap = nlx.lib.dll.nlx_proof_max_bytes(cds[0].handle)
    bufs = [np.zeros(cap, dtype=np.uint8) for _ in range(k)]
    jobs = (nlx.ProveJob * k)()
    for i in range(k):
        jobs[i].wires, jobs[i].public_inputs, jobs[i].proof_out, jobs[i].proof_cap = wires.data_ptr(), pis.ctypes.data, bufs[i].ctypes.data, cap
    handles = (ctypes.c_void_p * 3)(*[cd.handle for cd in cds])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rc = nlx.lib.dll.nlx_batch_prove(handles, 3, jobs, k)
    torch.cuda.synchronize()
    three = (time.perf_counter() - t0) / k * 1e3
    for cd in cds:
        cd.close()
    if rc != 0:
        raise RuntimeError("nlx_batch_prove failed with %d" % rc)
    return {"log_n_%d_ms_single_stream" % log_n: round(single, 2), "log_n_%d_ms_three_in_flight" % log_n: round(three, 2),
            "log_n_%d_proofs_per_s_three_in_flight" % log_n: round(1e3 / three, 1)}


def cpu_baseline_sync(args, nlx, st, sha256_result, sha512_result, ed25519_proof, gpu_seq_ms):
    """The oracle (C port, OpenMP) on a bounded sample of the same Sync step, on the GPU box's host cores:
    outer proof at 2^15 rows (scaled linearly to 2^log_n), the SHA-256 and SHA-512 STARKs at full size, the Ed25519 STARK
    at 2^5 slots (scaled linearly to the step's slots).  The same inputs are proved on the GPU and the BYTES compared."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py
    SA, SB, E = nlx.sha256_air, nlx.sha512_air, nlx.ed25519_air
    cores = min(len(os.sched_getaffinity(0)), 16)  # a one-GPU box grants 16 host cores
    os.environ["OMP_NUM_THREADS"] = str(cores)
    gate_mix = GATE_MIXES[args.gate_mix]
    ctx = st["ctxs"][3]
    parity = {}
    # outer: warm-up (spins up the OpenMP team, faults in the arenas), then the timed sample
    s_log = min(15, args.log_n)
    warm = nlx.SyntheticCircuit(max(s_log - 3, 6), seed=98, **gate_mix)
    wc = oracle_py.Circuit.from_synthetic(warm)
    wc.prove(warm.wires, warm.public_inputs)
    wc.close()
    syn = nlx.SyntheticCircuit(s_log, seed=99, **gate_mix)
    circ = oracle_py.Circuit.from_synthetic(syn)
    t = time.time()
    want = circ.prove(syn.wires, syn.public_inputs)
    t_outer = time.time() - t
    cd = nlx.CircuitData.from_synthetic(ctx, syn)
    got = cd.prove(syn.wires, syn.public_inputs)
    cd.close()
    parity["outer"] = {"log_n": s_log, "bytes_equal": got == want, "oracle_verifier_accepts": circ.verify(got) == 1}
    circ.close()
    scale_outer = 2.0 ** (args.log_n - s_log)
    # SHA-256 / SHA-512 STARKs of the step, full size: the oracle proves the reference trace of the same messages
    blocks, first, digest = SA.blocks_for_messages(st["sha_msgs"], st["lb256"])
    tr, _ = SA.reference_trace(blocks, first)
    t = time.time()
    tag = st["step_tag"]
    want = oracle_py.stark_prove_rounds(st["p256"].stark.desc, SA.cpu_rounds(blocks, first, tr), [int(v) for v in digest] + tag)
    t256 = time.time() - t
    parity["sha256"] = {"log_blocks": st["lb256"], "bytes_equal": want == sha256_result[0]}
    blocks, first, digest = SB.blocks_for_messages(st["sig_msgs"], st["lb512"])
    tr, _ = SB.reference_trace(blocks, first)
    t = time.time()
    want = oracle_py.stark_prove_rounds(st["p512"].stark.desc, SB.cpu_rounds(blocks, first, tr), [int(v) for v in SB.digest_halves(digest)] + tag)
    t512 = time.time() - t
    parity["sha512"] = {"log_blocks": st["lb512"], "bytes_equal": want == sha512_result[0]}
    # the relying party's side of the two bindings (post-timing; the oracle only parses the proofs' round values): the
    # SHA-512 proof's fingerprint covers every block and chaining value of the approvals' hashes, the Ed25519 proof's
    # covers (A, R, S, D, active) of every validator slot - with D taken from the SAME chaining values, so the digests
    # the curve equation used are the ones the SHA-512 proof is about
    v512 = oracle_py.stark_values(st["p512"].stark.desc, sha512_result[0])
    ved = oracle_py.stark_values(st["ped"].stark.desc, ed25519_proof)
    outs = SB.block_outputs(blocks, first)
    ends = [i for i in range(len(blocks)) if i + 1 == len(blocks) or first[i + 1]][-st["n_sigs"]:]   # filler messages come first
    tied = nlx.near_protocol.slots_with_digests(st["statement"], [outs[i] for i in ends])
    tied += [E.inactive_slot()] * ((1 << st["log_slots"]) - len(tied))
    v256 = oracle_py.stark_values(st["p256"].stark.desc, sha256_result[0])
    # values = public inputs (the AIR's own, then the 4-element step tag) | challenges | the fingerprint total
    parity["bindings"] = {"bytes_equal": tuple(v512[22:24]) == SB.fingerprint(blocks, first, v512[20:22])
                          and tuple(ved[8:10]) == E.fingerprint(tied, ved[6:8])
                          and list(v256[8:12]) == list(v512[16:20]) == list(ved[0:4]) == tag,
                          "note": "SHA-512 and Ed25519 round values recomputed from the step's public data; the Ed25519 slots' D = the "
                                  "SHA-512 proof's digests; the three transcripts open with the same step tag (hash of the 64 I/O bytes)"}
    # Ed25519: 2^5 slots of the step's signatures
    s_slots = min(5, st["log_slots"])
    pr2 = E.Ed25519Prover(ctx, s_slots)
    words = E.slots_to_words(st["bound_slots"][: 1 << s_slots])
    got = pr2.prove(words)
    host = pr2.generate_trace(words).cpu().numpy().view(np.uint64)
    tc = pr2.es.table_cols   # below 2^8 slots the 2^16-entry range table is spread over several periodic columns

    def cpu_round1(known):   # known = [alpha0, alpha1, gamma0, gamma1]
        acc, total_ = E.binding_columns(host, known[2:4])
        cols = np.concatenate([oracle_py.logup_round(host, E.LOOKUPS, 16, host[E.MULT:E.MULT + tc], known[:2], tc),
                               oracle_py.logup_round(host, E.LOOKUPS9, 9, host[E.MULT9], known[:2]), acc], axis=0)
        return cols, list(total_)
    t = time.time()
    want = oracle_py.stark_prove_rounds(pr2.stark.desc, lambda rnd, known: host if rnd == 0 else cpu_round1(known), [])
    t_ed = time.time() - t
    parity["ed25519"] = {"log_slots": s_slots, "bytes_equal": got == want}
    pr2.close()
    scale_ed = 2.0 ** (st["log_slots"] - s_slots)
    total = t_outer * scale_outer + t256 + t512 + t_ed * scale_ed
    base = {"value": 1.0 / total, "unit": "proofs/s", "cores": cores, "kind": "port",
            "sample": "oracle on one Sync step: outer proof at 2^%d rows in %.2f s (x%d, linear in rows), SHA-256 STARK 2^%d blocks "
                      "%.2f s, SHA-512 STARK 2^%d blocks %.2f s, Ed25519 STARK 2^%d slots %.2f s (x%d, linear in slots) -> %.1f s per "
                      "Sync proof; scalar C port, several times slower than plonky2's AVX2 prover would be: a baseline, not a target"
                      % (s_log, t_outer, int(scale_outer), st["lb256"], t256, st["lb512"], t512, s_slots, t_ed, int(scale_ed), total),
            "cpu_seconds_sampled": round(t_outer + t256 + t512 + t_ed, 2)}
    parity["all_bytes_equal"] = all(v["bytes_equal"] for v in parity.values())
    return base, parity


def run_ntt24(args, nlx, torch, rank, world, local, dist):
    """BASELINE.json configs[4]'s transform (the 2^24-point NTT behind the recursive wrap, here over Goldilocks): a batch of
    --ntt-cols columns x 2^--ntt-log-n points, resident in HBM, one forward NTT of the whole batch per step through
    nlx_ntt_batch (natural order in and out).  With N ranks the columns are split over the ranks - independent
    transforms, no collective (strong scaling); splitting ONE transform across GPUs (row-block four-step with an
    all-to-all) is out of scope."""
    import numpy as np
    if args.ntt_field == "bn254":
        return run_ntt24_bn254(args, nlx, torch, rank, world, local, dist)
    log_n, cols = args.ntt_log_n, args.ntt_cols
    if args.ntt_split and world > 1:
        return run_ntt24_split(args, nlx, torch, rank, world, local, dist)
    mine = [c for c in range(cols) if c % world == rank]
    n = 1 << log_n
    ctx = nlx.Context(local)
    g = torch.Generator(device="cpu").manual_seed(0x6E6C78 + rank)
    host = torch.randint(0, 2 ** 62, (max(len(mine), 1), n), generator=g, dtype=torch.int64)
    data = host.to("cuda:%d" % local)
    dll = nlx.lib.dll

    def step():
        if mine:
            ctx.check(dll.nlx_ntt_batch(ctx.handle, data.data_ptr(), len(mine), log_n, 0, 1))
    for _ in range(args.warmup):
        step()
    ctx.kernel_timing(True)
    barrier(dist, torch)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier(dist, torch)
    dt = reduce_max(dist, torch, time.perf_counter() - t0)
    kt, kr = ctx.kernel_stats("ntt_transform"), ctx.kernel_stats("ntt_reorder")
    ctx.kernel_timing(False)
    digest = None
    if world == 1 and log_n <= 20:     # one transform of the untouched input, hashed: what --ntt-split's reassembled result must equal
        import hashlib
        once = host.to(data.device)
        ctx.check(dll.nlx_ntt_batch(ctx.handle, once.data_ptr(), cols, log_n, 0, 1))
        digest = hashlib.sha256(once.cpu().numpy().tobytes()).hexdigest()
    out = None
    if rank == 0:
        alg = 16.0 * n * cols          # one read + one write of every element
        ach = (kt[2] / kt[0]) / (kt[1] / kt[0] * 1e-3) / 1e9 if kt[0] else 0.0
        out = {
            "metric": "NTT 2^%d x %d columns: transforms of the whole batch per second (BASELINE config 5's transform size)" % (log_n, cols),
            "value": args.steps / dt, "unit": "batch NTTs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u64 (Goldilocks field, integer)", "data": "synthetic",
            "config": {"result_sha256": digest,
                       "workload": "forward NTT of %d columns x 2^%d points (natural order in and out), columns resident in HBM, "
                                   "split over the ranks (no collective)" % (cols, log_n),
                       "columns_per_rank": len(mine), "whole_call_GBps_algorithmic": alg / (dt / args.steps) / 1e9,
                       "transform_ms_rank0": kt[1] / kt[0] if kt[0] else None, "reorder_ms_rank0": kr[1] / kr[0] if kr[0] else None,
                       "parallelism": "columns x%d" % world},
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                         "traffic": None, "kernel": "k_ntt_s<4> x 2 + k_ntt_c8_nat (ntt_transform: natural order in and out, no reordering pass)", "launches": kt[0],
                         "avg_launch_ms": kt[1] / kt[0] if kt[0] else None, "alg_bytes_per_launch": kt[2] / kt[0] if kt[0] else None,
                         "note": "a 2^24-point transform is three passes of eight levels (2^8 points per pass per workgroup), the last one writing "
                                 "every value at its natural position: actual traffic is 3 x the algorithmic 16 n bytes per column; the "
                                 "butterflies are integer-VALU work (DESIGN.md §4)"},
            "cpu_baseline": None,
        }
        if not args.no_cpu_baseline and world == 1:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import oracle_py
            cores = min(len(os.sched_getaffinity(0)), 16)
            s_log = min(log_n, 20)
            col = host[0, : 1 << s_log].numpy().view(np.uint64) % np.uint64(0xFFFFFFFF00000001)
            tc = time.time()
            ref = oracle_py.fft(col)
            dtc = time.time() - tc
            got = nlx.ntt(ctx, col.reshape(1, -1))[0]
            out["cpu_baseline"] = {"value": 1.0 / (dtc * cols * 2.0 ** (log_n - s_log) * (log_n / s_log)), "unit": "batch NTTs/s", "cores": 1,
                                   "kind": "port", "sample": "oracle radix-2 NTT of one 2^%d-point column in %.2f s on one core, scaled by "
                                   "n log n and the %d columns; GPU output of the same column equal: %s" % (s_log, dtc, cols, bool(np.array_equal(ref, got)))}
    ctx.close()
    return out


def run_ntt24_split(args, nlx, torch, rank, world, local, dist):
    """--workload ntt24 --ntt-split with N > 1 ranks (BASELINE.json configs[4] "split over GPUs"): every column's
    ONE 2^--ntt-log-n-point transform is split over the ranks - contiguous slices, log2 N pairwise exchanges of the slice over
    RCCL send / recv, then an independent transform per rank (near-light-client_amd/split_ntt.py).  Strong scaling of one
    transform; link-bound by construction.  At <= 2^20 points the reassembled result is hashed for comparison with the
    one-rank run."""
    import hashlib
    import numpy as np
    log_n, cols = args.ntt_log_n, args.ntt_cols
    n, m = 1 << log_n, (1 << log_n) // world
    ctx = nlx.Context(local)
    g = torch.Generator(device="cpu").manual_seed(0x6E6C78)     # the whole batch is the same for every world size
    host = torch.randint(0, 2 ** 62, (cols, n), generator=g, dtype=torch.int64)
    start = host[:, rank * m:(rank + 1) * m].contiguous().to("cuda:%d" % local)
    data = torch.empty_like(start)
    S = nlx.split_ntt

    def step():
        data.copy_(start)
        torch.cuda.synchronize()
        return S.split_ntt(ctx, data, log_n, rank, world, dist)
    for _ in range(args.warmup):
        step()
    barrier(dist, torch)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier(dist, torch)
    dt = reduce_max(dist, torch, time.perf_counter() - t0)
    digest = None
    if log_n <= 20:
        full = S.gather_natural(data, rank, world, dist)
        digest = hashlib.sha256(np.ascontiguousarray(full).tobytes()).hexdigest()
    ctx.close()
    if rank != 0:
        return None
    return {
        "metric": "Goldilocks NTT 2^%d x %d columns, every transform split over the ranks: transforms of the whole batch per second" % (log_n, cols),
        "value": args.steps / dt, "unit": "batch NTTs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "u64 (Goldilocks field, integer)", "data": "synthetic",
        "config": {"workload": "forward NTT of %d columns x 2^%d points, each transform split over %d ranks (contiguous slices, %d pairwise "
                               "slice exchanges, then 2^%d-point transforms per rank); output: rank r holds X[k], k = bitrev(r) mod %d"
                               % (cols, log_n, world, world.bit_length() - 1, log_n - world.bit_length() + 1, world),
                   "slice_bytes_per_rank": cols * m * 8, "bytes_sent_per_rank_per_step": cols * m * 8 * (world.bit_length() - 1),
                   "result_sha256": digest, "parallelism": "one transform x%d" % world},
        "roofline": None, "cpu_baseline": None,
    }


def run_ntt24_bn254(args, nlx, torch, rank, world, local, dist):
    """--workload ntt24 --ntt-field bn254: the transform of BASELINE.json configs[4] in its own field - a batch of columns
    x 2^--ntt-log-n points over BN254's scalar field (gnark-crypto fr.Element words, Montgomery form), resident in HBM,
    one forward NTT of the batch per step through nlx_bn254_ntt_batch; columns split over the ranks."""
    log_n, cols = args.ntt_log_n, args.ntt_cols
    mine = [c for c in range(cols) if c % world == rank]
    n = 1 << log_n
    ctx = nlx.Context(local)
    g = torch.Generator(device="cpu").manual_seed(0x626E + rank)
    host = torch.randint(0, 2 ** 60, (max(len(mine), 1), n, 4), generator=g, dtype=torch.int64)   # top word < 2^60: values < r
    data = host.to("cuda:%d" % local)
    dll = nlx.lib.dll
    order_flags = 1 | {"natural": 0, "dif": 2, "dit": 4}[args.ntt_order]   # fr.Element words; see nlx.h NLX_BN254_BITREV_*

    def step():
        if mine:
            ctx.check(dll.nlx_bn254_ntt_batch_coset(ctx.handle, data.data_ptr(), len(mine), log_n, 0, order_flags, None))
    for _ in range(args.warmup):
        step()
    ctx.kernel_timing(True)
    barrier(dist, torch)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier(dist, torch)
    dt = reduce_max(dist, torch, time.perf_counter() - t0)
    kt, kr = ctx.kernel_stats("bn254_ntt_transform"), ctx.kernel_stats("bn254_ntt_reorder")
    ctx.kernel_timing(False)
    out = None
    if rank == 0:
        ach = (kt[2] / kt[0]) / (kt[1] / kt[0] * 1e-3) / 1e9 if kt[0] else 0.0
        muls = n / 2 * log_n * len(mine)
        out = {
            "metric": "BN254 Fr NTT 2^%d x %d columns: transforms of the whole batch per second (the recursive wrap's transform, row f.4)" % (log_n, cols),
            "value": args.steps / dt, "unit": "batch NTTs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u256 (BN254 scalar field, Montgomery form, integer)", "data": "synthetic",
            "config": {"workload": "forward NTT of %d columns x 2^%d points over BN254 Fr (fr.Element words in and out, natural order), "
                                   "resident in HBM, columns split over the ranks (no collective)" % (cols, log_n),
                       "columns_per_rank": len(mine), "order": args.ntt_order, "transform_ms_rank0": kt[1] / kt[0] if kt[0] else None,
                       "reorder_ms_rank0": kr[1] / kr[0] if kr[0] else None,
                       "montgomery_multiplications_per_second_rank0": muls / (kt[1] / kt[0] * 1e-3) if kt[0] else None,
                       "parallelism": "columns x%d" % world},
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                         "traffic": None, "kernel": "k_bn_dif<3> x passes (bn254_ntt_transform)", "launches": kt[0],
                         "avg_launch_ms": kt[1] / kt[0] if kt[0] else None, "alg_bytes_per_launch": kt[2] / kt[0] if kt[0] else None,
                         "note": "64 bytes per element algorithmic (one read, one write); the transform makes log_n / 3 trips through "
                                 "HBM and is bound by the integer-VALU issue rate of the 256-bit Montgomery products"},
            "cpu_baseline": None,
        }
        if not args.no_cpu_baseline and world == 1:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import bn254_py
            s_log = min(log_n, 12)
            col = nlx.bn254_unpack(host[:1, : 1 << s_log].numpy().view("uint64"))[0]
            tc = time.time()
            ref = bn254_py.ntt([bn254_py.from_montgomery(x) for x in col])
            dtc = time.time() - tc
            got = nlx.bn254_unpack(nlx.bn254_ntt(ctx, host[:1, : 1 << s_log].numpy().view("uint64"), montgomery=True))[0]
            ok = [bn254_py.from_montgomery(x) for x in got] == ref
            out["cpu_baseline"] = {"value": 1.0 / (dtc * cols * 2.0 ** (log_n - s_log) * (log_n / s_log)), "unit": "batch NTTs/s", "cores": 1,
                                   "kind": "port", "sample": "pure-Python big-integer model, one 2^%d-point column in %.2f s, scaled by n log n and "
                                   "the %d columns (a model for parity, not a competitive CPU implementation); GPU output of the same column "
                                   "equal: %s" % (s_log, dtc, cols, ok)}
    ctx.close()
    return out


def run_plonk24(args, nlx, torch, rank, world, local, dist):
    """--workload plonk24: the PLONK prover's quotient chain of the recursive wrap (row f.4's third piece) on a coset of
    2^--ntt-log-n points (n = a quarter of that many gates): twelve FFTInverse(DIF) of n points, twelve FFT(DIT, OnCoset) of 4 n,
    the pointwise pass, one FFTInverse(OnCoset) of 4 n - then one KZG opening of the 3 n-coefficient quotient (evaluation +
    synthetic division as a scan; the MSM is msm24's job).  Inputs: uniform field elements resident in HBM (a random instance
    does not satisfy a circuit; the arithmetic is the same - small satisfying instances are the parity tests').  One rank."""
    import numpy as np
    log4 = args.ntt_log_n
    log_n = log4 - 2
    n = 1 << log_n
    ctx = nlx.Context(local)
    dev = "cuda:%d" % local
    g = torch.Generator(device="cpu").manual_seed(0x706C6B)

    def rand_fr(count):
        v = torch.randint(0, 2 ** 62, (count, 4), generator=g, dtype=torch.int64) * 4 + torch.randint(0, 4, (count, 4), generator=g, dtype=torch.int64)
        v[:, 3] = torch.randint(0, 0x30644e72e131a029, (count,), generator=g, dtype=torch.int64)
        return v.to(dev)
    names = ("ql", "qr", "qm", "qo", "qk", "s1", "s2", "s3", "l", "r", "o", "z")
    polys = {k: rand_fr(n) for k in names}
    out_t = torch.empty((3, n, 4), dtype=torch.int64, device=dev)
    sc = [5, 5, 25, 0x1234567, 0x89abcdef, 0x1357]   # any words below r: Montgomery forms of some field elements

    def step():
        nlx.bn254_plonk_quotient(ctx, polys, *sc, out=out_t)
    for _ in range(args.warmup):
        step()
    ctx.kernel_timing(True)
    barrier(dist, torch)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier(dist, torch)
    dt = time.perf_counter() - t0
    kq = ctx.kernel_stats("plonk_quotient")
    ctx.kernel_timing(False)
    # the opening of t (3 n coefficients) at a point: evaluation + quotient, no MSM
    flat = out_t.reshape(3 * n, 4)
    nlx.bn254_kzg_open(ctx, flat, 0x2468ace, want_quotient=False)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        nlx.bn254_kzg_open(ctx, flat, 0x2468ace, want_quotient=False)
    torch.cuda.synchronize()
    dt_open = (time.perf_counter() - t1) / args.steps
    ms_q = kq[1] / kq[0] if kq[0] else None
    alg = 32.0 * (4 * n) * (12 + 2 + 1)   # the pointwise pass: 12 evaluation columns + the two domain tables in, t out
    out = {
        "metric": "PLONK quotient chain over BN254 Fr on a coset of 2^%d points: chains per second (the recursive wrap's prover, row f.4)" % log4,
        "value": args.steps / dt, "unit": "chains/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "u256 (BN254 scalar field, Montgomery form, integer)", "data": "synthetic",
        "config": {"workload": "n = 2^%d gates: 12 x FFTInverse(DIF, n) + 12 x FFT(DIT, OnCoset, 4n) + pointwise quotient on 4n points + "
                               "FFTInverse(OnCoset, 4n), polynomials resident in HBM; domain tables (points, 1 / (n (x - 1)) by batch inversion) "
                               "rebuilt every call" % log_n,
                   "kzg_open_ms": dt_open * 1e3, "kzg_open_note": "evaluation + (p - p(zeta)) / (X - zeta) of the 3n-coefficient quotient as a "
                   "three-level Horner scan, without the MSM (msm24 times that)", "pointwise_kernel_ms": ms_q, "parallelism": "x1"},
        "roofline": {"bound": "hbm", "achieved": (alg / (ms_q * 1e-3) / 1e9) if ms_q else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": (alg / (ms_q * 1e-3) / 1e9 / HBM_PEAK_GBS) if ms_q else None, "traffic": None, "kernel": "k_plonk_quotient",
                     "launches": kq[0], "avg_launch_ms": ms_q, "alg_bytes_per_launch": alg,
                     "note": "32-byte elements; 15 columns of 4n per launch; ~33 256-bit products per point (~250 vector instructions each) make "
                             "it integer-VALU bound"},
        "cpu_baseline": None,
    }
    ctx.close()
    return out


def run_msm24_g2(args, nlx, torch, rank, world, local, dist):
    """--workload msm24 --msm-group g2: one BN254 G2 multi-scalar multiplication of 2^--ntt-log-n points per step (Groth16's B
    query; nlx_bn254_msm_g2, gnark-crypto G2Affine words).  The points are 1 024 distinct curve points tiled (made by the
    big-integer model: no device generator exists for G2), the scalars uniform below r; the result is checked against the
    model through the regrouped sum.  One rank."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import bn254_py   # input generation (points on the twist) and the post-timing check only
    log_n = args.ntt_log_n
    n, m = 1 << log_n, min(1024, 1 << log_n)     # the post-timing model check costs m G2 scalar multiplications (~30 ms each)
    rng = __import__("random").Random(98)
    acc, step_pt, base = bn254_py.g2_mul(rng.randrange(1, bn254_py.R), bn254_py.G2), bn254_py.g2_mul(rng.randrange(1, bn254_py.R), bn254_py.G2), []
    for _ in range(m):
        base.append(acc)
        acc = bn254_py.g2_add(acc, step_pt)
    ctx = nlx.Context(local)
    dev = "cuda:%d" % local
    pts = torch.from_numpy(nlx.bn254_g2_pack(base).view(np.int64)).to(dev).repeat(n // m, 1).contiguous()
    g = torch.Generator(device="cpu").manual_seed(0x6D736E)
    ks = torch.randint(0, 2 ** 62, (n, 4), generator=g, dtype=torch.int64) * 4 + torch.randint(0, 4, (n, 4), generator=g, dtype=torch.int64)
    ks[:, 3] = torch.randint(0, 0x30644e72e131a029, (n,), generator=g, dtype=torch.int64)
    d_ks = ks.to(dev)
    for _ in range(args.warmup):
        nlx.bn254_msm_g2(ctx, pts, d_ks)
    ctx.kernel_timing(True)
    barrier(dist, torch)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = nlx.bn254_msm_g2(ctx, pts, d_ks)
    barrier(dist, torch)
    dt = time.perf_counter() - t0
    kt = ctx.kernel_stats("bn254_msm_g2")
    ctx.kernel_timing(False)
    ms = kt[1] / kt[0] if kt[0] else None
    out = {
        "metric": "BN254 G2 MSM of 2^%d points: multi-scalar multiplications per second (Groth16's B query, row f.4)" % log_n,
        "value": args.steps / dt, "unit": "MSMs/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "u256 x 2 (BN254 base field's quadratic extension, integer)", "data": "synthetic",
        "config": {"workload": "one G2 MSM of 2^%d points x 254-bit scalars (gnark-crypto G2Affine / fr.Element words), resident in HBM" % log_n,
                   "distinct_points": m, "device_ms_rank0": ms, "parallelism": "x1"},
        "roofline": {"bound": "hbm", "achieved": (160.0 * n / (ms * 1e-3) / 1e9) if ms else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": (160.0 * n / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if ms else None, "traffic": None, "kernel": "bn254_msm_g2",
                     "launches": kt[0], "avg_launch_ms": ms, "alg_bytes_per_launch": 160.0 * n,
                     "note": "160 bytes per point algorithmic; integer-VALU bound (see roofline_valu): three base-field products per Fq2 product"},
        "roofline_valu": {"bound": "integer VALU issue", "achieved": (16.0 * n / (ms * 1e-3) / 1e9) if ms else None,
                          "peak": 1024 * 64 / (3.7 * 3400 * 1.99), "unit": "G bucket additions/s",
                          "frac": (16.0 * n / (ms * 1e-3) / 1e9) / (1024 * 64 / (3.7 * 3400 * 1.99)) if ms else None,
                          "peak_model": "the G1 model x 3.7 (Karatsuba Fq2 products, every result tightened: DESIGN.md §13)"},
        "cpu_baseline": None,
    }
    if not args.no_cpu_baseline:
        kw = ks.numpy().view(np.uint64).astype(object)
        ints = kw[:, 0] + (kw[:, 1] << 64) + (kw[:, 2] << 128) + (kw[:, 3] << 192)
        tc = time.time()
        want = bn254_py.msm_g2([int(sum(ints[j::m])) % bn254_py.R for j in range(m)], base)
        per_mul = (time.time() - tc) / m
        out["cpu_baseline"] = {"value": 1.0 / (per_mul * n), "unit": "MSMs/s", "cores": 1, "kind": "port",
                               "sample": "pure-Python big-integer model: %d G2 scalar multiplications at %.1f ms each (a model for parity, not a "
                                         "competitive CPU implementation); GPU result equal to the model's regrouped sum: %s"
                                         % (m, per_mul * 1e3, nlx.bn254_g2_unpack(res) == want)}
    ctx.close()
    return out


def run_msm24(args, nlx, torch, rank, world, local, dist):
    """--workload msm24: the KZG commitment of the recursive wrap (BASELINE.json configs[4]'s size): one
    BN254 G1 multi-scalar multiplication of 2^--ntt-log-n points per step through nlx_bn254_msm_g1, points and scalars
    resident in HBM in gnark-crypto's layouts.  The points are 2^--ntt-log-n DISTINCT curve points - (i + 1) P, made on the
    GPU by nlx_bn254_g1_multiples, so the bucket kernel's gathers go to HBM as they would with a real SRS -, the scalars
    uniform below r.  With N ranks every rank owns a slice of the points (the 64-byte partial results are gathered and added
    on rank 0: one G1 addition per rank, no data-path collective)."""
    import hashlib
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import bn254_py   # input generation (points on the curve) and the post-timing check only
    log_n = args.ntt_log_n
    n = (1 << log_n) // world
    rng = __import__("random").Random(99)
    base = bn254_py.g1_mul(rng.randrange(1, bn254_py.R), bn254_py.G1)
    ctx = nlx.Context(local)
    dev = "cuda:%d" % local
    # point i of the job = (i + 1) base; this rank's slice starts at rank * n: multiples of base from ((rank n + 1) base) on
    pts = nlx.bn254_g1_multiples(ctx, base, n * world, device=dev)[rank * n:(rank + 1) * n].contiguous() if world > 1 else \
        nlx.bn254_g1_multiples(ctx, base, n, device=dev)
    # the job's scalars are the same for every world size (each rank takes its slice), so that the joined result can be compared
    g = torch.Generator(device="cpu").manual_seed(0x6D736D)
    n_all = n * world
    ks = torch.randint(0, 2 ** 62, (n_all, 4), generator=g, dtype=torch.int64) * 4 + torch.randint(0, 4, (n_all, 4), generator=g, dtype=torch.int64)
    ks[:, 3] = torch.randint(0, 0x30644e72e131a029, (n_all,), generator=g, dtype=torch.int64)   # top word below r's: uniform scalars < r, canonical form
    ks_all = ks
    ks = ks[rank * n:(rank + 1) * n].contiguous()
    d_ks = ks.to(dev)

    def step():
        return nlx.bn254_msm_g1(ctx, pts, d_ks)
    for _ in range(args.warmup):
        step()
    ctx.kernel_timing(True)
    barrier(dist, torch)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    barrier(dist, torch)
    dt = reduce_max(dist, torch, time.perf_counter() - t0)
    kt = ctx.kernel_stats("bn254_msm_g1")
    ctx.kernel_timing(False)
    if world > 1:   # the ranks' partial results meet on rank 0: 64 bytes each, one G1 addition per rank
        mine_t = torch.from_numpy(res.view(np.int64).copy()).to("cpu" if dist.get_backend() == "gloo" else dev)
        parts = [torch.empty_like(mine_t) for _ in range(world)]
        dist.all_gather(parts, mine_t)
        res = nlx.bn254_g1_sum(np.stack([p.cpu().numpy().view(np.uint64) for p in parts]))
    out = None
    if rank == 0:
        ms = kt[1] / kt[0] if kt[0] else None
        adds_per_s = 16.0 * n / (ms * 1e-3) if ms else None       # one mixed addition per (point, window) pair
        out = {
            "metric": "BN254 G1 MSM of 2^%d points: multi-scalar multiplications per second (the recursive wrap's KZG commitment, row f.4)" % log_n,
            "value": args.steps / dt, "unit": "MSMs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u256 (BN254 base field, Montgomery form, integer)", "data": "synthetic",
            "config": {"workload": "one G1 MSM of 2^%d points x 254-bit scalars (gnark-crypto G1Affine / fr.Element words), resident "
                                   "in HBM, points split over the ranks" % log_n,
                       "points_per_rank": n, "distinct_points": n * world, "device_ms_rank0": ms,
                       "result_sha256": hashlib.sha256(np.ascontiguousarray(res, dtype=np.uint64).tobytes()).hexdigest(),
                       "points_per_second": world * n * args.steps / dt,
                       "bucket_additions_per_second_rank0": adds_per_s, "parallelism": "points x%d" % world},
            "roofline": {"bound": "hbm", "achieved": (96.0 * n / (ms * 1e-3) / 1e9) if ms else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (96.0 * n / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if ms else None, "traffic": None,
                         "kernel": "bn254_msm_g1 (digits, scan, 16 radix sorts, bucket sums, window reduction)", "launches": kt[0],
                         "avg_launch_ms": ms, "alg_bytes_per_launch": 96.0 * n,
                         "note": "96 bytes per point algorithmic (point + scalar once); the job is bound by the integer-VALU issue rate - "
                                 "see roofline_valu"},
            # the bound that binds: one mixed Jacobian addition per (point, window) pair, ~3 400 vector instructions each
            # (1 644 of them multiply-adds), every instruction at the measured 1.99 ns per wave-instruction per SIMD
            "roofline_valu": {"bound": "integer VALU issue", "achieved": adds_per_s / 1e9 if adds_per_s else None,
                              "peak": 1024 * 64 / (3400 * 1.99), "unit": "G bucket additions/s",
                              "frac": (adds_per_s / 1e9) / (1024 * 64 / (3400 * 1.99)) if adds_per_s else None,
                              "peak_model": "1 024 SIMDs x 64 lanes / (3 400 instructions per mixed addition x 1.99 ns, profiles/r02_valu_ubench_v5.txt); "
                                            "sorting and the window reduction are inside the measured time"},
            "cpu_baseline": None,
        }
        if world > 1:
            # the joined result against the model: sum_i k_i (i + 1) base over ALL ranks' scalars, one scalar multiplication
            kw = ks_all.numpy().view(np.uint64).astype(object)
            ints = kw[:, 0] + (kw[:, 1] << 64) + (kw[:, 2] << 128) + (kw[:, 3] << 192)
            total = int((ints * np.arange(1, n * world + 1, dtype=object)).sum()) % bn254_py.R
            out["config"]["joined_result_equals_model"] = bool(nlx.bn254_g1_unpack(res) == bn254_py.g1_mul(total, base))
        if not args.no_cpu_baseline and world == 1:
            # the model on the same input: sum_i k_i (i + 1) base = (sum_i k_i (i + 1) mod r) base - one scalar multiplication
            # pins the whole result; its speed is measured on a sample of plain scalar multiplications
            kw = ks.numpy().view(np.uint64).astype(object)
            ints = kw[:, 0] + (kw[:, 1] << 64) + (kw[:, 2] << 128) + (kw[:, 3] << 192)
            total = int((ints * np.arange(1, n + 1, dtype=object)).sum()) % bn254_py.R
            ok = nlx.bn254_g1_unpack(res) == bn254_py.g1_mul(total, base)
            tc, sample = time.time(), 2000
            for i in range(sample):
                bn254_py.g1_mul(int(ints[i]), base)
            per_mul = (time.time() - tc) / sample
            out["cpu_baseline"] = {"value": 1.0 / (per_mul * n), "unit": "MSMs/s", "cores": 1, "kind": "port",
                                   "sample": "pure-Python big-integer model: %d scalar multiplications at %.2f ms each; a term-by-term MSM of "
                                             "all 2^%d points would take %.0f s (a model for parity, not a competitive CPU implementation); GPU "
                                             "result equal to the model's (sum k_i (i + 1)) P: %s" % (sample, per_mul * 1e3, log_n, per_mul * n, ok)}
    ctx.close()
    return out


def run_verify128(args, nlx, torch, rank, world, local, dist, t_outer_2p15=None):
    """BASELINE.json configs[3]: the VerifyCircuit 128 x 4 map-reduce job (nearx/src/verify.rs:69-90), sharded over the ranks -
    the STRONG-scaling workload (README.md:123: "No parallelisation", 22 s per batch on the reference's CPU).  The CPU leg
    times the test oracle on one reduce-sized proof and (unless the Sync leg already did) one 2^15-row proof and scales
    linearly in rows: 32 map + 31 reduce + 1 outer proofs."""
    from importlib import import_module
    mr = import_module("nlx_amd.mapreduce")
    verify_outer = cpu = None
    if not args.no_cpu_baseline and rank == 0:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))

        def verify_outer(syn, proof):   # the checker (test oracle's plonky2 verifier) on the job's outer proof
            import oracle_py
            oc = oracle_py.Circuit.from_synthetic(syn)
            ok = oc.verify(proof) == 1
            oc.close()
            return ok

        def cpu():
            import oracle_py
            cores = min(len(os.sched_getaffinity(0)), 16)
            os.environ["OMP_NUM_THREADS"] = str(cores)
            mix = dict(pct_poseidon=30, pct_arithmetic=30, pct_base_sum=5, pct_constant=5)   # GpuTreeProver's gate mix

            def timed(log_n, seed):
                syn = nlx.SyntheticCircuit(log_n, seed=seed, num_public_inputs=8, **mix)
                oc = oracle_py.Circuit.from_synthetic(syn)
                t = time.time()
                oc.prove(syn.wires, syn.public_inputs)
                dt_ = time.time() - t
                oc.close()
                return dt_
            s_map = min(15, args.map_log_n)
            timed(max(args.reduce_log_n - 3, 6), 97)     # spins up the OpenMP team
            t_red = timed(args.reduce_log_n, 98)
            t_map = t_outer_2p15 if (t_outer_2p15 is not None and s_map == 15) else timed(s_map, 99)
            total = 32 * t_map * 2.0 ** (args.map_log_n - s_map) + 32 * t_red
            return {"value": 1.0 / total, "unit": "proofs/s", "cores": cores, "kind": "port",
                    "sample": "oracle prover: one reduce-sized proof (2^%d rows) %.2f s x 32, one 2^%d-row proof %.2f s x %d (linear in rows) "
                              "x 32 map proofs -> %.0f s per 128 x 4 job" % (args.reduce_log_n, t_red, s_map, t_map,
                                                                            int(2.0 ** (args.map_log_n - s_map)), total),
                    "external_anchor": "the reference's README.md:123 quotes ~12 min per 128 x 4 job = 0.0014 proofs/s (22 s per "
                                       "batch of 4, no parallelisation) on a Ryzen 9 7950X, witness generation and RPC included - "
                                       "not like-for-like"}
    return mr.bench_verify128(args, nlx, torch, rank, world, local, dist, verify_outer, cpu)


def relaunch_under_torchrun(args):
    """--gpus N > 1 without a launcher (WORLD_SIZE unset): start the N ranks ourselves, as CHILD processes of a process that
    has not touched the GPU yet (no exec: the pool forbids replacing a process that initialised HIP, and this one has not even
    imported torch), and leave with the launcher's exit code.  A WORLD_SIZE that disagrees with --gpus is an error, not a
    warning: the line would otherwise report n_gpus = WORLD_SIZE for a run the caller believes to be on N GPUs."""
    world = os.environ.get("WORLD_SIZE")
    if world is not None:
        if int(world) != args.gpus:
            print("error: --gpus %d but WORLD_SIZE=%s; launch with python -m torch.distributed.run --nnodes=1 --nproc-per-node %d "
                  "--master-addr 127.0.0.1 bench.py --gpus %d ..." % (args.gpus, world, args.gpus, args.gpus), file=sys.stderr)
            sys.exit(2)
        return
    if args.gpus <= 1:
        return
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: --gpus %d without a launcher, starting %s" % (args.gpus, " ".join(cmd[1:9])), file=sys.stderr)
    sys.exit(subprocess.call(cmd, env=env))


def main():
    args = parse()
    relaunch_under_torchrun(args)   # before anything touches the GPU
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (there is no CPU fallback)")
    rank, world, local, dist = dist_setup(args.gpus)
    import nlxpkg
    nlx = nlxpkg.load()
    if args.log_n is None:
        args.log_n = 18 if args.workload == "sync" else 16
    if args.workload == "sync":
        out = run_sync(args, nlx, torch, rank, world, local, dist)
    elif args.workload == "outer":
        out = run_outer(args, nlx, torch, rank, world, local, dist)
    elif args.workload == "ntt24":
        out = run_ntt24(args, nlx, torch, rank, world, local, dist)
    elif args.workload == "msm24":
        out = (run_msm24_g2 if args.msm_group == "g2" else run_msm24)(args, nlx, torch, rank, world, local, dist)
    elif args.workload == "plonk24":
        out = run_plonk24(args, nlx, torch, rank, world, local, dist)
    elif args.workload == "stark":
        out = run_stark(args, nlx, torch, rank, world, local, dist)
    elif args.workload == "ed25519":
        out = run_ed25519(args, nlx, torch, rank, world, local, dist)
    elif args.workload in ("sha256", "sha512"):
        out = run_sha256(args, nlx, torch, rank, world, local, dist)
    else:
        out = run_verify128(args, nlx, torch, rank, world, local, dist)
    if rank == 0:
        if os.environ.get("NLX_BENCH_REHEARSAL") == "1":
            out["config"]["rehearsal"] = "all ranks share GPU 0, gloo backend: NOT a scaling measurement"
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
