#!/bin/bash
# PMC traffic of this round's NEW kernels (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 runs, kernel trace only):
#   the outer proof at 2^18 rows with a 2^16-pair lookup table (k_lookup_terms, k_lk_*), the Sync step's SHA-512 STARK
#   (k_hash_lde_groups) and the PLONK quotient chain on a coset of 2^22 points (k_plonk_quotient, k_plonk_domain)
# -> gpurun_out/r03/pmc_new/r03_pmc_new_kernels.txt
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT="$ROOT/gpurun_out/r03/pmc_new"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {  # name, counter, bench args...
  local name=$1 ctr=$2; shift 2
  rocprofv3 --pmc $ctr --kernel-trace -d "$OUT/${name}_$ctr" -o p --output-format csv -- python3 "$ROOT/bench.py" "$@" > /dev/null 2>&1
}
for C in FETCH_SIZE WRITE_SIZE; do
  run lookup $C --workload outer --log-n 18 --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline --lookup-tables 1 --lookup-bits 16 --lookups 200000
  run sha512 $C --workload sha512 --log-blocks 7 --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline
  run plonk $C --workload plonk24 --ntt-log-n 22 --steps 2 --warmup 1
done
cd "$ROOT"
python3 - "$OUT" <<'PY' > "$OUT/r03_pmc_new_kernels.txt"
import csv, sys, collections
root = sys.argv[1]
WANT = {"lookup": ("k_lookup_terms", "k_lk_scan", "k_lk_row_terms", "k_lk_count", "k_lk_write"),
        "sha512": ("k_hash_lde_groups", "k_hash_lde_leaves", "k_air_quotient"),
        "plonk": ("k_plonk_quotient", "k_plonk_domain", "k_horner_local", "k_horner_final", "k_bn_dit", "k_bn_dif3")}
print("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, kernel trace only; KB x 1024; gfx950's FETCH_SIZE counts half of coalesced")
print("streaming reads, so fetched = 2 x FETCH_SIZE).  Per kernel: launches in the run, bytes per launch.")
for name, kernels in WANT.items():
    tot = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for i, c in enumerate(("FETCH_SIZE", "WRITE_SIZE")):
        try:
            rows = list(csv.DictReader(open("%s/%s_%s/p_counter_collection.csv" % (root, name, c))))
        except FileNotFoundError:
            continue
        for r in rows:
            if r["Counter_Name"] != c:
                continue
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
            for wnt in kernels:
                if wnt in k:
                    tot[wnt][i] += float(r["Counter_Value"])
                    if i == 0:
                        tot[wnt][2] += 1
    print("\n== %s" % name)
    for k in kernels:
        f, w, n = tot[k]
        if n:
            print("%-22s launches %4d   fetched %9.3f MB   written %9.3f MB   per launch" % (k, n, 2 * f * 1024 / 1e6 / n, w * 1024 / 1e6 / n))
PY
cat "$OUT/r03_pmc_new_kernels.txt"
