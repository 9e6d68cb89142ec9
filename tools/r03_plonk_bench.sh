#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p "$ROOT/gpurun_out/r03"
cd "$ROOT"
for L in 20 24; do
  timeout -k 10 500 python3 bench.py --workload plonk24 --ntt-log-n $L --steps 3 --warmup 1 > gpurun_out/r03/plonk_$L.json 2> gpurun_out/r03/plonk_$L.err || { tail -5 gpurun_out/r03/plonk_$L.err; exit 1; }
  python3 -c "
import json; d = json.load(open('gpurun_out/r03/plonk_$L.json')); print($L, 'ms_per_step', d['ms_per_step'], d['config']['pointwise_kernel_ms'], 'kzg_open_ms', d['config']['kzg_open_ms'], d['roofline']['frac'])"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$ROOT/gpurun_out/r03/plonk_prof" -o p --output-format csv -- python3 "$ROOT/bench.py" --workload plonk24 --ntt-log-n 24 --steps 2 --warmup 1 > /dev/null 2>&1
cd "$ROOT"
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r03/plonk_prof/**/p_kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
for r in rows[:16]:
    print("%-60s %6s %10.3f %10.1f" % (r["Name"].replace("(anonymous namespace)::", "").split("(")[0][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
