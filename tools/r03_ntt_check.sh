#!/bin/bash
# round 3: parity tests that exercise the NTT kernels, then the two timing lines (ntt24, outer 2^18 one proof at a time)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-new}
mkdir -p "$ROOT/gpurun_out/r03"
cd "$ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_prover.py -m gpu -x -q > gpurun_out/r03/tests_$TAG.txt 2>&1
echo "tests rc=$?" >> gpurun_out/r03/tests_$TAG.txt
tail -5 gpurun_out/r03/tests_$TAG.txt
timeout -k 10 300 python bench.py --workload ntt24 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r03/ntt24_$TAG.json 2> gpurun_out/r03/ntt24_$TAG.err
python3 -c "
import json;d=json.load(open('gpurun_out/r03/ntt24_$TAG.json'));print('ntt24', d['ms_per_step'], d.get('kernel_ms_per_step'))"
timeout -k 10 300 python bench.py --workload outer --log-n 18 --steps 4 --warmup 1 --inflight 1 --no-cpu-baseline > gpurun_out/r03/outer18_$TAG.json 2> gpurun_out/r03/outer18_$TAG.err
python3 -c "
import json;d=json.load(open('gpurun_out/r03/outer18_$TAG.json'));print('outer18', d['ms_per_step'], d.get('kernel_ms_per_step'), d.get('stage_ms_last_proof'))"
