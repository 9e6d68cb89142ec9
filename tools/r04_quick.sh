#!/bin/bash
# round 4: Poseidon micro-benchmark + primitive / prover parity + outer 2^18 one proof at a time;  bash tools/r04_quick.sh <tag> [extra pytest files]
ROOT=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-x}; shift
mkdir -p "$ROOT/gpurun_out/r04"
cd "$ROOT"
timeout -k 10 200 tools/ubench/pub > gpurun_out/r04/ubench_$TAG.txt 2>&1
echo "ubench rc=$?" >> gpurun_out/r04/ubench_$TAG.txt
grep -E "poseidon v|==|differ" gpurun_out/r04/ubench_$TAG.txt
timeout -k 10 900 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_prover.py "$@" -m gpu -x -q > gpurun_out/r04/tests_$TAG.txt 2>&1
echo "tests rc=$?" >> gpurun_out/r04/tests_$TAG.txt
tail -4 gpurun_out/r04/tests_$TAG.txt
for LN in 18; do
timeout -k 10 300 python bench.py --workload outer --log-n $LN --steps 4 --warmup 1 --inflight 1 --no-cpu-baseline > gpurun_out/r04/outer${LN}_$TAG.json 2> gpurun_out/r04/outer${LN}_$TAG.err
python3 -c "
import json;d=json.load(open('gpurun_out/r04/outer${LN}_$TAG.json'));print('outer$LN', d['ms_per_step'], d.get('kernel_ms_per_proof'))"
done
