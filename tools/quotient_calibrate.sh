#!/bin/bash
# Real per-item cost of k_quotient's work items (19 gates in gate-list order, then the permutation argument of each challenge):
# a calibration build (NLX_BUILD_VARIANT=qcal NLX_EXTRA_FLAGS=-DNLX_QUOTIENT_CALIBRATE python near-light-client_amd/build.py)
# lets every wave evaluate ONE item; the quotient_eval stage time then ranks the items.  gpurun -- 'bash tools/quotient_calibrate.sh'
cd "${GRAFT_REPO_ROOT:-$PWD}"
for k in $(seq 0 20); do
  NLX_Q_ONLY=$k NLX_BUILD_VARIANT=qcal python bench.py --workload outer --steps 3 --warmup 1 --inflight 1 --no-cpu-baseline 2>/dev/null | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('item %2d  quotient_eval %.3f ms' % ($k, d['stage_ms_last_proof']['quotient_eval']))"
done
