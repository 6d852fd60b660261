#!/bin/bash
O=gpurun_out/r03
mkdir -p $O
ENF_HIP_LIB=$PWD/variants/libenf_a3.so timeout -k 10 600 python -m pytest tests/test_gpu_forward.py tests/test_gpu_golden.py -m gpu -x -q 2>&1 | tail -3
AB_ROUNDS=3 timeout -k 10 600 python scripts/ab_kernels.py - variants/libenf_a3.so 2>&1 | tee $O/c12_ab.log
