#!/bin/bash
# round 3: rocprofv3 kernel stats + SQ counters + FETCH/WRITE of the NTT workloads after the tile kernels
ROOT=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-v1}
mkdir -p "$ROOT/gpurun_out/r03"
SQ="SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"
cd "$ROOT"
bash tools/pmc_sq.sh ntt24_$TAG "$SQ" --workload ntt24 --steps 3 --warmup 1 --no-cpu-baseline
bash tools/pmc_sq.sh outer18_$TAG "$SQ" --workload outer --log-n 18 --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$ROOT/gpurun_out/r03/stats_ntt24_$TAG" -o s --output-format csv -- python3 "$ROOT/bench.py" --workload ntt24 --steps 3 --warmup 1 --no-cpu-baseline > "$ROOT/gpurun_out/r03/ntt24_prof_$TAG.json" 2>/dev/null
rocprofv3 --kernel-trace --stats -d "$ROOT/gpurun_out/r03/stats_outer18_$TAG" -o s --output-format csv -- python3 "$ROOT/bench.py" --workload outer --log-n 18 --steps 4 --warmup 1 --inflight 1 --no-cpu-baseline > "$ROOT/gpurun_out/r03/outer18_prof_$TAG.json" 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d "$ROOT/gpurun_out/r03/pmc_ntt24_${TAG}_$c" -o p --output-format csv -- python3 "$ROOT/bench.py" --workload ntt24 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
done
cd "$ROOT"
head -8 gpurun_out/r03/stats_ntt24_$TAG/s_kernel_stats.csv | cut -c1-160
head -12 gpurun_out/r03/stats_outer18_$TAG/s_kernel_stats.csv | cut -c1-160
python3 - "$TAG" <<'PY'
import csv, sys, collections
tag = sys.argv[1]
tot = collections.defaultdict(lambda: [0.0, 0.0, 0])
for i, c in enumerate(("FETCH_SIZE", "WRITE_SIZE")):
    for r in csv.DictReader(open("gpurun_out/r03/pmc_ntt24_%s_%s/p_counter_collection.csv" % (tag, c))):
        if r["Counter_Name"] == c:
            k = r["Kernel_Name"].split("(")[0][:60]
            tot[k][i] += float(r["Counter_Value"])
            if i == 0:
                tot[k][2] += 1
with open("gpurun_out/r03/pmc_traffic_ntt24_%s.txt" % tag, "w") as f:
    f.write("per kernel name, summed over the run's launches: FETCH_SIZE x 2 (gfx950: the counter sees half of streaming reads) and WRITE_SIZE, in GB\n")
    for k, (fe, w, n) in sorted(tot.items(), key=lambda kv: -(2 * kv[1][0] + kv[1][1]))[:8]:
        f.write("%-62s launches %4d  fetched %8.3f GB  written %8.3f GB\n" % (k, n, 2 * fe * 1024 / 1e9, w * 1024 / 1e9))
print(open("gpurun_out/r03/pmc_traffic_ntt24_%s.txt" % tag).read())
PY
