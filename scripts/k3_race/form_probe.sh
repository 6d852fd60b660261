#!/bin/bash
# Packed-operand forms in the store_probe context (scripts/k3_race/README.md, "Packed-operand forms"): one run per build.
# usage (GPU box): bash scripts/k3_race/form_probe.sh [ITERS]   -> gpurun_out/r03/form_probe.log
IT=${1:-4000}
O=gpurun_out/r03; mkdir -p $O
LOG=$O/form_probe.log
: >> $LOG; echo "#### $(date -u) $(hostname)" >> $LOG
# positive controls first: `pre` = K3 exactly as it was before the fix (commit 1b5f394's enf_pair_bwd.hip linked into today's library),
# `old` = today's K3 with the plain C++ apply loop (ENF_LN_APPLY_ASM=0, SLP-packed into the failing form).  If neither deviates on
# this box, the box does not show the effect and the forms are not run (a clean result would mean nothing).
CTRL=0
for v in pre old; do
  export ENF_HIP_LIB=$PWD/variants/libenf_$v.so
  echo "== control $v" | tee -a $LOG
  POLLUTE=1 timeout -k 10 600 python scripts/k3_race/store_probe.py 6000 128 1 bf16 2>&1 | grep -v "^it \|signature\|amdgpu.ids" | tail -2 | tee $O/.ctrl | tee -a $LOG
  n=$(grep -o "differ from the first [0-9]*" $O/.ctrl | grep -o "[0-9]*$")
  CTRL=$((CTRL + ${n:-0}))
done
if [ "$CTRL" = "0" ]; then echo "controls clean on this box: forms not run" | tee -a $LOG; exit 0; fi
for v in - f1p0 f1p1 f2p0 f3p0 f4p0 f5p0 f6p0 f7p0; do
  if [ "$v" = "-" ]; then unset ENF_HIP_LIB; else export ENF_HIP_LIB=$PWD/variants/libenf_$v.so; fi
  echo "== $v" | tee -a $LOG
  # the form computes the LayerNorm it stands for: one parity case through the model API (f32-accurate reference, bf16 kernels)
  timeout -k 10 200 python -m pytest tests/test_gpu_backward.py -m gpu -x -q -k "test_backward_shapes and bf16 and 128-1" 2>&1 | tail -1 | tee -a $LOG
  POLLUTE=1 timeout -k 10 400 python scripts/k3_race/store_probe.py $IT 128 1 bf16 2>&1 | grep -v "^it \|signature\|amdgpu.ids" | tail -2 | tee -a $LOG
done
