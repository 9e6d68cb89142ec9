#!/bin/bash
# the Ed25519 STARK at the Sync step's size (2^7 slots), one proof at a time: stage times + kernel trace
ROOT=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p "$ROOT/gpurun_out/r03"
cd "$ROOT"
python3 bench.py --workload ed25519 --log-slots 7 --steps 5 --warmup 2 --inflight 1 --no-cpu-baseline > gpurun_out/r03/ed7.json 2> gpurun_out/r03/ed7.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$ROOT/gpurun_out/r03/ed7_prof" -o p --output-format csv -- python3 "$ROOT/bench.py" --workload ed25519 --log-slots 7 --steps 4 --warmup 1 --inflight 1 --no-cpu-baseline > /dev/null 2>&1
cd "$ROOT"
python3 - <<'PY'
import csv, glob, json
d = json.load(open("gpurun_out/r03/ed7.json"))
print("ms_per_step", d["ms_per_step"]); print({k: v for k, v in d.items() if "stage" in k or "kernel" in k})
print(json.dumps(d["config"])[:600])
f = glob.glob("gpurun_out/r03/ed7_prof/**/p_kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot / 1e6)
for r in rows[:24]:
    print("%-60s %6s %10.3f %10.1f" % (r["Name"].replace("(anonymous namespace)::", "").split("(")[0][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
