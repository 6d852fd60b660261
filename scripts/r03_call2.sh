#!/bin/bash
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_backward.py -m gpu -x -q > $O/c2_tests_spec.log 2>&1; echo "spec tests rc=$?"; tail -2 $O/c2_tests_spec.log
ENF_HIP_LIB=variants/libenf_dnlps.so timeout -k 10 500 python -m pytest tests/test_gpu_backward.py -m gpu -x -q -k "not duplicate" > $O/c2_tests_dnlps.log 2>&1; echo "dnlps tests rc=$?"; tail -2 $O/c2_tests_dnlps.log
timeout -k 10 600 python scripts/ab_kernels.py variants/libenf_nospec.so - variants/libenf_dnl.so variants/libenf_dnlprio.so variants/libenf_dnlps.so 2>&1 | tee $O/c2_ab.log
