#!/bin/bash
# round-3 GPU call 1: new parity tests, K3 split / priority A/B, phase stamps
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 420 python -m pytest tests/test_gpu_backward.py tests/test_gpu_golden.py -m gpu -x -q -k "bench_latent_count or cfg3" > $O/c1_tests.log 2>&1; echo "tests rc=$?" | tee -a $O/c1_tests.log
tail -3 $O/c1_tests.log
timeout -k 10 600 python scripts/ab_kernels.py - variants/libenf_ns2.so variants/libenf_ns4.so variants/libenf_ns2nr.so variants/libenf_prio.so variants/libenf_sprio.so 2>&1 | tee $O/c1_ab.log
ENF_HIP_LIB=variants/libenf_st1.so timeout -k 10 120 python scripts/stamps_k3.py > $O/c1_stamps_ns1.log 2>&1
ENF_HIP_LIB=variants/libenf_st2.so timeout -k 10 120 python scripts/stamps_k3.py > $O/c1_stamps_ns2.log 2>&1
tail -12 $O/c1_stamps_ns1.log
