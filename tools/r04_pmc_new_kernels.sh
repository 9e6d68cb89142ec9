#!/bin/bash
# round 4: PMC passes of this round's NEW kernels (separate rocprofv3 runs, kernel trace only):
#   ntt24    k_ntt_s<4> and k_ntt_c8_nat: FETCH_SIZE, WRITE_SIZE, SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES
#   sync     the generated AIR kernels (airgen_*::kernel) and the batched leaf hashing: the same counters
# -> gpurun_out/r04/pmc_new/r04_pmc_new_kernels.txt
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT="$ROOT/gpurun_out/r04/pmc_new"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {  # name, tag, counters..., -- bench args
  local name=$1 tag=$2; shift 2
  local ctrs=()
  while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
  shift
  rocprofv3 --pmc "${ctrs[@]}" --kernel-trace -d "$OUT/${name}_$tag" -o p --output-format csv -- python3 "$ROOT/bench.py" "$@" > /dev/null 2>&1
}
NTT="--workload ntt24 --steps 2 --warmup 1 --no-cpu-baseline"
SYNC="--steps 2 --warmup 1 --no-cpu-baseline --no-extra"
run ntt fetch FETCH_SIZE -- $NTT
run ntt write WRITE_SIZE -- $NTT
run ntt insts SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES -- $NTT
run sync fetch FETCH_SIZE -- $SYNC
run sync write WRITE_SIZE -- $SYNC
run sync insts SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVES -- $SYNC
cd "$ROOT"
python3 - "$OUT" <<'PY' > "$OUT/r04_pmc_new_kernels.txt"
import csv, sys, collections
root = sys.argv[1]
WANT = {"ntt": ("k_ntt_s", "k_ntt_c8_nat"), "sync": ("airgen_", "k_hash_lde_leaves", "k_air_combine", "k_merkle_fused", "k_merkle_level")}
print("rocprofv3 --pmc (separate passes, kernel trace only).  FETCH_SIZE / WRITE_SIZE are KB; gfx950's FETCH_SIZE counts half of coalesced")
print("streaming reads, so fetched = 2 x FETCH_SIZE.  Per kernel name: launches in the run, per-launch averages.")
for name, kernels in WANT.items():
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(int)
    for tag in ("fetch", "write", "insts"):
        try:
            rows = list(csv.DictReader(open("%s/%s_%s/p_counter_collection.csv" % (root, name, tag))))
        except FileNotFoundError:
            continue
        for r in rows:
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
            if not any(w in k for w in kernels):
                continue
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] in ("FETCH_SIZE",):
                cnt[k] += 1
    print("\n== %s" % name)
    for k in sorted(tot, key=lambda kk: -tot[kk].get("FETCH_SIZE", 0)):
        n = max(cnt[k], 1)
        t = tot[k]
        line = "%-70s launches %4d  fetched %8.3f GB  written %8.3f GB per launch" % (k[:70], n, 2 * t.get("FETCH_SIZE", 0) * 1024 / 1e9 / n, t.get("WRITE_SIZE", 0) * 1024 / 1e9 / n)
        if "SQ_INSTS_VALU" in t:
            line += "  VALU %.3e  waves %.3e  VALU/wave %.0f" % (t["SQ_INSTS_VALU"] / n, t.get("SQ_WAVES", 0) / n, t["SQ_INSTS_VALU"] / max(t.get("SQ_WAVES", 1), 1))
        if "SQ_INSTS_LDS" in t:
            line += "  LDS insts/wave %.0f  bank-conflict cycles/wave %.0f" % (t["SQ_INSTS_LDS"] / max(t.get("SQ_WAVES", 1), 1), t.get("SQ_LDS_BANK_CONFLICT", 0) / max(t.get("SQ_WAVES", 1), 1))
        print(line)
PY
cat "$OUT/r04_pmc_new_kernels.txt"
