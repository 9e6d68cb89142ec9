#!/bin/bash
# round 4: batched commitment rounds - parity, then the Sync line with and without;  bash tools/r04_batches.sh <tag>
ROOT=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-x}
mkdir -p "$ROOT/gpurun_out/r04"
cd "$ROOT"
timeout -k 10 700 python -m pytest tests/test_gpu_stark.py tests/test_gpu_primitives.py tests/test_gpu_airgen.py tests/test_sha256_air.py tests/test_sha512_air.py -m gpu -x -q > gpurun_out/r04/batches_tests_$TAG.txt 2>&1
echo "tests rc=$?" >> gpurun_out/r04/batches_tests_$TAG.txt
tail -5 gpurun_out/r04/batches_tests_$TAG.txt
for B in 512 0 256 1024; do
timeout -k 10 400 python bench.py --no-extra --no-cpu-baseline --stark-batch-cols $B > gpurun_out/r04/bench_sync_b${B}_$TAG.json 2> gpurun_out/r04/bench_sync_b${B}_$TAG.err
echo "bench B=$B rc=$?"
python3 -c "
import json;d=json.load(open('gpurun_out/r04/bench_sync_b${B}_$TAG.json'))
print('B=$B value', round(d['value'],2), 'ms', round(d['ms_per_step'],2), 'one at a time', d['ms_one_proof_at_a_time'], 'floor', d['config']['outer_rows_floor_from_stark_verification'])
print('   kernel ms', d['kernel_ms_per_step'])
"
done
