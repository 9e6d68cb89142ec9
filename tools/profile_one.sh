#!/bin/bash
# rocprofv3 kernel statistics of one bench.py command on the GPU box:  bash tools/profile_one.sh <name> <bench.py args...>
# -> gpurun_out/r02/<name>.json (the bench line) and gpurun_out/r02/stats_<name>/s_kernel_stats.csv, top kernels printed
NAME=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p "$ROOT/gpurun_out/r02"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$ROOT/gpurun_out/r02/stats_$NAME" -o s --output-format csv -- python3 "$ROOT/bench.py" "$@" > "$ROOT/gpurun_out/r02/$NAME.json" 2>/dev/null
cd "$ROOT"
python3 - "$NAME" <<'PY'
import json, csv, sys
name = sys.argv[1]
d = json.load(open("gpurun_out/r02/%s.json" % name))
print(d["ms_per_step"], d["config"].get("trace_gen_ms_per_step"), d.get("stage_ms_last_proof"))
for r in list(csv.DictReader(open("gpurun_out/r02/stats_%s/s_kernel_stats.csv" % name)))[:16]:
    print("%-50s calls %4s avg %9.1f us %6s%%" % (r["Name"][:50], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
