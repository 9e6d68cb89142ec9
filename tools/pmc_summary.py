"""Summarise the rocprofv3 --pmc passes collected by tools/collect_profiles_r02.sh (pmc_fetch / pmc_write / pmc_insts under
the given directory) for `bench.py --workload outer --steps 2 --warmup 1 --inflight 1` at 2^16 rows: per kernel of interest
the LAST proof's launches, FETCH_SIZE / WRITE_SIZE in KB as reported, HBM traffic = 2 x FETCH_SIZE + WRITE_SIZE (gfx950's
FETCH_SIZE counts half of a coalesced streaming read: MI355X_MICROARCH.md, HBM section), and the algorithmic bytes of
SURVEY.md §8(d) beside it.   python tools/pmc_summary.py gpurun_out/profiles_r02_<tag> [log_n]   (log_n = the rows of the
outer proof the passes ran on: 16 by default, 18 = the headline's size, tools/r03_pmc_outer18.sh)"""
import collections
import csv
import json
import os
import sys

LOG_N, RATE_BITS = (int(sys.argv[2]) if len(sys.argv) > 2 else 16), 3
L = 1 << (LOG_N + RATE_BITS)
N_CS, N_W, N_ZS, NC = 85, 135, 20, 2  # constants+sigmas (3 selectors + 2 + 80), wires, Z + partial products, challenges


def load(path):
    rows = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            rows[(r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Counter_Name"])].append(
                (int(r["Dispatch_Id"]), float(r["Counter_Value"]), int(r["Grid_Size"])))
    return rows


def last(rows, kernel, counter, n):
    v = sorted(rows.get((kernel, counter), []))
    return v[-n:] if v else []


def main(root):
    fetch = load(os.path.join(root, "pmc_fetch", "p_counter_collection.csv"))
    write = load(os.path.join(root, "pmc_write", "p_counter_collection.csv"))
    insts = load(os.path.join(root, "pmc_insts", "p_counter_collection.csv"))
    out = {"log_n": LOG_N, "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVES (separate passes), "
                                     "bench.py --workload outer --log-n %d --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline, MI355X" % LOG_N,
           "units": "FETCH_SIZE / WRITE_SIZE are KB; traffic_bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 FETCH_SIZE counts 1/2 of coalesced streaming reads)"}
    # k_quotient: one launch per proof
    q = "nlx::k_quotient"
    f, w = last(fetch, q, "FETCH_SIZE", 1), last(write, q, "WRITE_SIZE", 1)
    if f and w:
        alg = 8 * L * (N_CS + N_W + N_ZS + NC) + 8 * L * NC   # every column once + z(gx) + the two outputs
        traffic = (2 * f[0][1] + w[0][1]) * 1024
        d = {"FETCH_SIZE_KB": f[0][1], "WRITE_SIZE_KB": w[0][1], "traffic_bytes": traffic, "algorithmic_bytes": alg, "ratio": traffic / alg,
             "fetched_bytes": 2 * f[0][1] * 1024, "fetched_over_algorithmic": 2 * f[0][1] * 1024 / alg}
        for c in ("SQ_INSTS_VALU", "SQ_INSTS_VMEM_RD", "SQ_WAVES"):
            v = last(insts, q, c, 1)
            if v:
                d[c] = v[0][1]
        if "SQ_INSTS_VALU" in d:
            d["valu_wave_instructions_per_tile_of_64_points"] = d["SQ_INSTS_VALU"] / (L / 64)
            d["vmem_rd_wave_instructions_per_tile_of_64_points"] = d.get("SQ_INSTS_VMEM_RD", 0) / (L / 64)
        out["quotient_19_gates"] = d
        out["quotient_traffic_over_algorithmic"] = traffic / alg            # reads + writes (the writes include spilled registers)
        out["quotient_fetch_over_algorithmic"] = 2 * f[0][1] * 1024 / alg   # reads alone
    # k_hash_lde_leaves: three launches per proof (wires 135, zs 20, quotient 16 columns)
    h = "nlx::k_hash_lde_leaves"
    f, w = last(fetch, h, "FETCH_SIZE", 3), last(write, h, "WRITE_SIZE", 3)
    if len(f) == 3 and len(w) == 3:
        tot_t = tot_a = 0.0
        per = []
        for cols, (_, fk, _), (_, wk, _) in zip((N_W, N_ZS, 16), f, w):
            alg = 8 * cols * L + 32 * L
            traffic = (2 * fk + wk) * 1024
            per.append({"cols": cols, "FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk, "traffic_bytes": traffic, "algorithmic_bytes": alg, "ratio": traffic / alg})
            tot_t += traffic
            tot_a += alg
        out["hash_lde_leaves_last_proof"] = per
        out["hash_lde_leaves_fetch_over_algorithmic"] = tot_t / tot_a
        v = last(insts, h, "SQ_INSTS_VALU", 3)
        if v:
            out["hash_lde_leaves_valu_wave_instructions_per_permutation_wires_launch"] = v[0][1] / (L / 64 * 17)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
