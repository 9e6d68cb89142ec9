#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p "$ROOT/gpurun_out/r03"
cd "$ROOT"
timeout -k 10 900 python -m pytest tests/test_gpu_prover.py -m gpu -x -q 2>&1 | tail -3 &&
for L in 16 18; do
python3 bench.py --workload outer --log-n $L --steps 6 --warmup 2 --inflight 1 --no-cpu-baseline > gpurun_out/r03/outer_q_$L.json 2>/dev/null
python3 -c "
import json; d = json.load(open('gpurun_out/r03/outer_q_$L.json')); print($L, 'ms', d['ms_per_step'], d['kernel_ms_per_proof'])"
done
