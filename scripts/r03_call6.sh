#!/bin/bash
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_weight_grads.py tests/test_gpu_narrow.py -m gpu -x -q > $O/c6_wgrad.log 2>&1; echo "wgrad rc=$?"; tail -15 $O/c6_wgrad.log
bash scripts/k3_race/replay_ab.sh 6000 > /dev/null 2>&1; cat $O/replay_ab.log
