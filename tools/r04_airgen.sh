#!/bin/bash
# round 4: generated AIR kernels against the interpreter, then the Sync line;  bash tools/r04_airgen.sh <tag>
ROOT=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-x}
mkdir -p "$ROOT/gpurun_out/r04"
cd "$ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_airgen.py tests/test_sha256_air.py tests/test_sha512_air.py tests/test_ed25519_air.py -m gpu -x -q > gpurun_out/r04/airgen_tests_$TAG.txt 2>&1
echo "tests rc=$?" >> gpurun_out/r04/airgen_tests_$TAG.txt
tail -5 gpurun_out/r04/airgen_tests_$TAG.txt
timeout -k 10 400 python bench.py --no-extra --no-cpu-baseline > gpurun_out/r04/bench_sync_noextra_$TAG.json 2> gpurun_out/r04/bench_sync_noextra_$TAG.err
echo "bench rc=$?"
python3 -c "
import json;d=json.load(open('gpurun_out/r04/bench_sync_noextra_$TAG.json'))
print('value', d['value'], 'ms', d['ms_per_step'])
print('one at a time', d['ms_one_proof_at_a_time'])
print('kernel ms', d['kernel_ms_per_step'])
"
NLX_AIR_VM=1 timeout -k 10 400 python bench.py --no-extra --no-cpu-baseline > gpurun_out/r04/bench_sync_noextra_vm_$TAG.json 2> gpurun_out/r04/bench_sync_noextra_vm_$TAG.err
python3 -c "
import json;d=json.load(open('gpurun_out/r04/bench_sync_noextra_vm_$TAG.json'))
print('VM: value', d['value'], 'ms', d['ms_per_step'])
print('VM: kernel ms', d['kernel_ms_per_step'])
"
