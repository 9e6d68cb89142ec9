#!/bin/bash
# Times the MSM (2^24 and 2^20 points) for the kernel-tuning builds made with NLX_BUILD_VARIANT (near-light-client_amd/build.py):
#   gpurun -- 'bash tools/msm_variants.sh msmw3 msmw4'
cd "${GRAFT_REPO_ROOT:-$PWD}"
mkdir -p gpurun_out/r02
python -m pytest tests/test_gpu_bn254.py -x -q -m gpu -k msm 2>&1 | tail -2
for v in "" "$@"; do
  for ln in 20 24; do
    NLX_BUILD_VARIANT=$v timeout -k 10 300 python bench.py --workload msm24 --ntt-log-n $ln --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-8s 2^%d  %.2f ms' % ('$v' or 'default', $ln, d['config']['device_ms_rank0']))"
  done
done
