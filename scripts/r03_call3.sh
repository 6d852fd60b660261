#!/bin/bash
O=gpurun_out/r03
mkdir -p $O
AB_ROUNDS=3 timeout -k 10 900 python scripts/ab_kernels.py - variants/libenf_b0.so variants/libenf_edy.so variants/libenf_lna2.so variants/libenf_f2.so variants/libenf_all.so 2>&1 | tee $O/c3_ab.log
ENF_HIP_LIB=variants/libenf_all.so timeout -k 10 300 python -m pytest tests/test_gpu_backward.py -m gpu -x -q -k "not duplicate" > $O/c3_tests_all.log 2>&1; echo "all tests rc=$?"; tail -2 $O/c3_tests_all.log
