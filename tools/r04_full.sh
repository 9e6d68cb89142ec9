#!/bin/bash
# round 4: the whole GPU suite, then the default bench line;  bash tools/r04_full.sh <tag>
ROOT=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-x}
mkdir -p "$ROOT/gpurun_out/r04"
cd "$ROOT"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04/gpu_suite_$TAG.txt 2>&1
echo "tests rc=$?" >> gpurun_out/r04/gpu_suite_$TAG.txt
tail -5 gpurun_out/r04/gpu_suite_$TAG.txt
timeout -k 10 400 python bench.py > gpurun_out/r04/bench_sync_$TAG.json 2> gpurun_out/r04/bench_sync_$TAG.err
echo "bench rc=$?"
python3 -c "
import json;d=json.load(open('gpurun_out/r04/bench_sync_$TAG.json'))
print('value', d['value'], 'ms', d['ms_per_step'], d['config']['stark_variant'], d['config']['outer_rows_floor_from_stark_verification'])
print('one at a time', d['ms_one_proof_at_a_time'])
print('kernel ms', d['kernel_ms_per_step'])
print('roofline', d['roofline'])
print('verify128', {k: d['verify128'].get(k) for k in ('proofs_per_s','ms_per_job','map_starks_ms_per_job','level_ms')})
print('parity', d.get('parity_checked',{}).get('all_bytes_equal'))
"
