#!/bin/bash
# FETCH_SIZE / WRITE_SIZE (separate rocprofv3 --pmc passes, kernel trace only) of one bench.py command, summed per kernel name:
#   bash tools/pmc_one.sh <name> <bench.py args...>   -> gpurun_out/r02/pmc_<name>.txt
NAME=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p "$ROOT/gpurun_out/r02"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d "$ROOT/gpurun_out/r02/pmc_${NAME}_$c" -o p --output-format csv -- python3 "$ROOT/bench.py" "$@" > /dev/null 2>&1
done
cd "$ROOT"
python3 - "$NAME" <<'PY' > "gpurun_out/r02/pmc_$NAME.txt"
import csv, sys, collections
name = sys.argv[1]
tot = collections.defaultdict(lambda: [0.0, 0.0, 0])
for i, c in enumerate(("FETCH_SIZE", "WRITE_SIZE")):
    for r in csv.DictReader(open("gpurun_out/r02/pmc_%s_%s/p_counter_collection.csv" % (name, c))):
        if r["Counter_Name"] == c:
            k = r["Kernel_Name"].split("(")[0][:60]
            tot[k][i] += float(r["Counter_Value"])
            if i == 0:
                tot[k][2] += 1
print("per kernel name, summed over the run's launches: FETCH_SIZE x 2 (gfx950: the counter sees half of streaming reads) and WRITE_SIZE, in GB")
for k, (f, w, n) in sorted(tot.items(), key=lambda kv: -(2 * kv[1][0] + kv[1][1]))[:12]:
    print("%-62s launches %4d  fetched %8.3f GB  written %8.3f GB" % (k, n, 2 * f * 1024 / 1e9, w * 1024 / 1e9))
PY
cat "gpurun_out/r02/pmc_$NAME.txt"
