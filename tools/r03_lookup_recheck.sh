#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p "$ROOT/gpurun_out/r03/pmc_lk2"
cd "$ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_lookup.py tests/test_c_example.py -m gpu -x -q 2>&1 | tail -2 || exit 1
python3 bench.py --workload outer --log-n 18 --steps 6 --warmup 2 --inflight 1 --no-cpu-baseline --lookup-tables 1 --lookup-bits 16 --lookups 200000 2>/dev/null | python3 -c "
import sys, json; d = json.loads(sys.stdin.read()); print('with table: ms', round(d['ms_per_step'], 2), d['kernel_ms_per_proof'])"
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --pmc $C --kernel-trace -d "$ROOT/gpurun_out/r03/pmc_lk2/$C" -o p --output-format csv -- python3 "$ROOT/bench.py" --workload outer --log-n 18 --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline --lookup-tables 1 --lookup-bits 16 --lookups 200000 > /dev/null 2>&1
done
cd "$ROOT"
python3 - <<'PY'
import csv
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    tot, n = 0.0, 0
    for r in csv.DictReader(open("gpurun_out/r03/pmc_lk2/%s/p_counter_collection.csv" % c)):
        if r["Counter_Name"] == c and "k_lookup_terms" in r["Kernel_Name"]:
            tot += float(r["Counter_Value"]); n += 1
    print("k_lookup_terms", c, "per launch MB:", (2 if c == "FETCH_SIZE" else 1) * tot * 1024 / 1e6 / max(n, 1), "launches", n)
PY
