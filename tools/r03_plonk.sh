#!/bin/bash
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_gpu_bn254_plonk.py -m gpu -x -q > gpurun_out/r03/tests_plonk.txt 2>&1
echo "rc=$?" >> gpurun_out/r03/tests_plonk.txt
tail -30 gpurun_out/r03/tests_plonk.txt
