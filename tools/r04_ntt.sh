#!/bin/bash
# round 4: NTT parity + the 2^24 x 16 workload with and without the reordering pass;  bash tools/r04_ntt.sh <tag>
ROOT=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-x}
mkdir -p "$ROOT/gpurun_out/r04"
cd "$ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_ntt_tiles.py -m gpu -x -q -k "ntt" > gpurun_out/r04/ntt_tests_$TAG.txt 2>&1
echo "tests rc=$?" >> gpurun_out/r04/ntt_tests_$TAG.txt
tail -4 gpurun_out/r04/ntt_tests_$TAG.txt
timeout -k 10 300 python bench.py --workload ntt24 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r04/ntt24_$TAG.json 2> gpurun_out/r04/ntt24_$TAG.err
NLX_NTT_REORDER=1 timeout -k 10 300 python bench.py --workload ntt24 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r04/ntt24_reorder_$TAG.json 2> gpurun_out/r04/ntt24_reorder_$TAG.err
python3 -c "
import json
for f in ('ntt24_$TAG','ntt24_reorder_$TAG'):
    d=json.load(open('gpurun_out/r04/%s.json'%f)); print(f, d['ms_per_step'], d['config']['transform_ms_rank0'], d['config']['reorder_ms_rank0'], d['config']['whole_call_GBps_algorithmic'])
"
