#!/bin/bash
# full GPU suite under the other candidate generator pair (libnlx_gen2021.so / liboracle_gen2021.so, built by __graft_entry__.build())
mkdir -p gpurun_out/r03
NLX_GL_GENERATOR_SET=2021 timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r03/tests_gpu_final_gen2021.txt 2>&1
echo "rc=$?" >> gpurun_out/r03/tests_gpu_final_gen2021.txt
tail -4 gpurun_out/r03/tests_gpu_final_gen2021.txt
