#!/bin/bash
# cost of the lookup argument at the headline's size: outer proof 2^18 rows, with / without one 2^16-pair table and 200 000
# lookups, one proof at a time; then the kernel trace of the lookup run.  -> gpurun_out/r03/lookup_*.{json,txt}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p "$ROOT/gpurun_out/r03"
cd "$ROOT"
python3 bench.py --workload outer --log-n 18 --steps 6 --warmup 2 --inflight 1 --no-cpu-baseline > gpurun_out/r03/lookup_off.json 2> gpurun_out/r03/lookup_off.err &&
python3 bench.py --workload outer --log-n 18 --steps 6 --warmup 2 --inflight 1 --no-cpu-baseline --lookup-tables 1 --lookup-bits 16 --lookups 200000 > gpurun_out/r03/lookup_on.json 2> gpurun_out/r03/lookup_on.err &&
cd /tmp && export TMPDIR=/tmp &&
rocprofv3 --kernel-trace --stats -d "$ROOT/gpurun_out/r03/lookup_prof" -o p --output-format csv -- python3 "$ROOT/bench.py" --workload outer --log-n 18 --steps 4 --warmup 1 --inflight 1 --no-cpu-baseline --lookup-tables 1 --lookup-bits 16 --lookups 200000 > /dev/null 2>&1
cd "$ROOT"
python3 - <<'PY' > gpurun_out/r03/lookup_kernels.txt
import csv, glob
f = glob.glob("gpurun_out/r03/lookup_prof/**/p_kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
print("rocprofv3 --kernel-trace --stats: outer 2^18 rows + one 2^16-pair table, 200 000 lookups; 5 proofs (1 warm-up + 4)")
print("%-70s %6s %12s %10s" % ("kernel", "calls", "total_ms", "avg_us"))
for r in rows[:16]:
    print("%-70s %6s %12.3f %10.1f" % (r["Name"].replace("(anonymous namespace)::", "").split("(")[0][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
print()
for r in rows:
    if "lk_" in r["Name"] or "lookup" in r["Name"]:
        print("%-70s %6s %12.3f %10.1f" % (r["Name"].replace("(anonymous namespace)::", "").split("(")[0][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
python3 - <<'PY'
import json
for k in ("off", "on"):
    d = json.load(open("gpurun_out/r03/lookup_%s.json" % k))
    print(k, "ms_per_step %.2f" % d["ms_per_step"], d["kernel_ms_per_proof"], d["stage_ms_last_proof"])
PY
cat gpurun_out/r03/lookup_kernels.txt
