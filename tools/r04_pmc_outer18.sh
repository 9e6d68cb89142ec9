#!/bin/bash
# round 4: the three PMC passes (FETCH_SIZE / WRITE_SIZE / SQ instruction counts, separate rocprofv3 runs, kernel trace only)
# of the outer proof at the HEADLINE's 2^18 rows -> gpurun_out/r04/pmc18_<tag>/r04_pmc_traffic_outer_2p18.json
# (the ratio bench.py's roofline.traffic cites) and the per-kernel byte table
ROOT=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-v1}
OUT="$ROOT/gpurun_out/r04/pmc18_$TAG"
mkdir -p "$OUT"
ARGS="--workload outer --log-n 18 --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/pmc_fetch" -o p --output-format csv -- python3 "$ROOT/bench.py" $ARGS > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/pmc_write" -o p --output-format csv -- python3 "$ROOT/bench.py" $ARGS > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVES --kernel-trace -d "$OUT/pmc_insts" -o p --output-format csv -- python3 "$ROOT/bench.py" $ARGS > /dev/null 2>&1
cd "$ROOT"
python3 tools/pmc_summary.py "$OUT" 18 > "$OUT/r04_pmc_traffic_outer_2p18.json"
python3 - "$OUT" <<'PY' > "$OUT/r04_pmc_by_kernel_outer_2p18.txt"
import csv, sys, collections
root = sys.argv[1]
tot = collections.defaultdict(lambda: [0.0, 0.0, 0])
for i, (d, c) in enumerate((("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE"))):
    for r in csv.DictReader(open("%s/%s/p_counter_collection.csv" % (root, d))):
        if r["Counter_Name"] == c:
            k = r["Kernel_Name"].split("(")[0][:60]
            tot[k][i] += float(r["Counter_Value"])
            if i == 0:
                tot[k][2] += 1
print("per kernel name, summed over the run's launches (3 proofs): FETCH_SIZE x 2 (gfx950: the counter sees half of streaming reads) and WRITE_SIZE, in GB")
for k, (f, w, n) in sorted(tot.items(), key=lambda kv: -(2 * kv[1][0] + kv[1][1]))[:14]:
    print("%-62s launches %4d  fetched %8.3f GB  written %8.3f GB" % (k, n, 2 * f * 1024 / 1e9, w * 1024 / 1e9))
PY
cat "$OUT/r04_pmc_by_kernel_outer_2p18.txt"
python3 -c "
import json;d=json.load(open('$OUT/r04_pmc_traffic_outer_2p18.json'));print({k:v for k,v in d.items() if 'over_algorithmic' in k or 'per_' in k}); print(d.get('quotient_19_gates'))"
