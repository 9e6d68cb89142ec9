#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p "$ROOT/gpurun_out/r03"
cd "$ROOT"
timeout -k 10 600 python -m pytest tests/test_ed25519_air.py -m gpu -x -q 2>&1 | tail -3 &&
python3 bench.py --workload ed25519 --log-slots 7 --steps 5 --warmup 2 --inflight 1 --no-cpu-baseline > gpurun_out/r03/ed7b.json 2> gpurun_out/r03/ed7b.err
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r03/ed7b.json"))
print("ms_per_step", d["ms_per_step"]); print({k: v for k, v in d.items() if "stage" in k or "kernel" in k})
PY
