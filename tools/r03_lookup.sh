#!/bin/bash
# lookup-argument parity on the GPU box:  bash tools/r03_lookup.sh  -> gpurun_out/r03/tests_lookup.txt
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_gpu_lookup.py -m gpu -x -q > gpurun_out/r03/tests_lookup.txt 2>&1
echo "rc=$?" >> gpurun_out/r03/tests_lookup.txt
tail -30 gpurun_out/r03/tests_lookup.txt
