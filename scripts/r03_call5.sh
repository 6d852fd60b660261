#!/bin/bash
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > $O/c5_gpu_suite.log 2>&1; echo "suite rc=$?"; tail -3 $O/c5_gpu_suite.log
bash scripts/k3_race/form_probe.sh 4000 > /dev/null 2>&1; tail -30 $O/form_probe.log
ENF_HIP_LIB=variants/libenf_st3.so timeout -k 10 120 python scripts/stamps_k3.py > $O/c5_stamps.log 2>&1; tail -12 $O/c5_stamps.log
