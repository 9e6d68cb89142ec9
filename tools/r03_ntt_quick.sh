#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p "$ROOT/gpurun_out/r03"
cd "$ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_ntt_tiles.py tests/test_gpu_primitives.py -m gpu -x -q 2>&1 | tail -2 &&
python3 bench.py --workload ntt24 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r03/ntt24_q.json 2>/dev/null
python3 -c "
import json; d = json.load(open('gpurun_out/r03/ntt24_q.json')); print('ntt24 ms', d['ms_per_step'], json.dumps(d['config'])[:400])"
