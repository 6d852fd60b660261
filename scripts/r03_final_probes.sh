#!/bin/bash
# run-to-run probes of round 2's K3 investigation on the round's FINAL kernels (new packed instruction forms in K2 / K3 / tail):
# store_probe = K3 with the activation store + K4, bitwise against the first call, behind the polluter kernel; fwd_probe = K2
O=gpurun_out/r03
mkdir -p $O
POLLUTE=1 timeout -k 10 400 python scripts/k3_race/store_probe.py 6000 > $O/final_store_probe_128_1.log 2>&1; tail -2 $O/final_store_probe_128_1.log
POLLUTE=1 timeout -k 10 400 python scripts/k3_race/store_probe.py 3000 128 2 bf16 > $O/final_store_probe_128_2.log 2>&1; tail -2 $O/final_store_probe_128_2.log
POLLUTE=1 timeout -k 10 300 python scripts/k3_race/fwd_probe.py 3000 128 2 bf16 > $O/final_fwd_probe.log 2>&1; tail -2 $O/final_fwd_probe.log
POLLUTE=1 timeout -k 10 300 python scripts/k3_race/train_probe.py 1500 > $O/final_train_probe.log 2>&1; tail -2 $O/final_train_probe.log
