#!/bin/bash
# SQ counters (one rocprofv3 --pmc pass, kernel trace only) of one bench.py command, summed per kernel name:
#   bash tools/pmc_sq.sh <name> "<counter> <counter> ..." <bench.py args...>   -> gpurun_out/r03/sq_<name>.txt
# At most 8 SQ counters per pass (MI355X_MICROARCH.md, rocprofv3 PMC slots).
NAME=$1; shift
COUNTERS=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p "$ROOT/gpurun_out/r03"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $COUNTERS --kernel-trace -d "$ROOT/gpurun_out/r03/sq_${NAME}" -o p --output-format csv -- python3 "$ROOT/bench.py" "$@" > /dev/null 2>&1
cd "$ROOT"
python3 - "$NAME" "$COUNTERS" <<'PY' > "gpurun_out/r03/sq_$NAME.txt"
import csv, sys, collections
name, counters = sys.argv[1], sys.argv[2].split()
tot = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(set)
for r in csv.DictReader(open("gpurun_out/r03/sq_%s/p_counter_collection.csv" % name)):
    k = r["Kernel_Name"].split("(")[0][:60]
    tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
    calls[k].add(r["Dispatch_Id"])
print("per kernel name, summed over the run's launches (SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count quad-cycles per wave)")
print("%-60s %6s " % ("kernel", "calls") + " ".join("%22s" % c for c in counters))
for k, d in sorted(tot.items(), key=lambda kv: -kv[1].get(counters[0], 0))[:14]:
    print("%-60s %6d " % (k, len(calls[k])) + " ".join("%22.0f" % d.get(c, 0) for c in counters))
PY
cat "gpurun_out/r03/sq_$NAME.txt"
