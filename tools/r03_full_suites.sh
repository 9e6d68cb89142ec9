#!/bin/bash
# full GPU suite under both generator pairs  -> gpurun_out/r03/tests_gpu_final*.txt
mkdir -p gpurun_out/r03
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r03/tests_gpu_final.txt 2>&1
echo "rc=$?" >> gpurun_out/r03/tests_gpu_final.txt
tail -4 gpurun_out/r03/tests_gpu_final.txt
