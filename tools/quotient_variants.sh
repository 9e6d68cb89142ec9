#!/bin/bash
# Times k_quotient (stage quotient_eval of a 2^16-row, nineteen-gate proof, one proof at a time) for the kernel-tuning builds
# made with NLX_BUILD_VARIANT (see near-light-client_amd/build.py):  gpurun -- 'bash tools/quotient_variants.sh q8w2 q4w2 ...'
cd "${GRAFT_REPO_ROOT:-$PWD}"
for v in "" "$@"; do
  NLX_BUILD_VARIANT=$v python bench.py --workload outer --steps 6 --warmup 2 --inflight 1 --no-cpu-baseline ${NLX_BENCH_ARGS} 2>/dev/null | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-10s quotient_eval %.3f ms   proof %.2f ms' % ('$v' or 'default', d['stage_ms_last_proof']['quotient_eval'], d['ms_per_step']))"
done
