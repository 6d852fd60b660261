#!/bin/bash
# Hazard or instruction?  The pre-fix K3 (commit 1b5f394's enf_pair_bwd.hip) rebuilt from its device assembly three ways
# (scripts/k3_race/asm_replay.sh; variants/libenf_pre{0,A,B}.so): 0 = unedited, A = `s_nop 3` in front of every
# `v_pk_add_f32 .. op_sel:[0,1] neg` (640 places), B = each of them replaced by two v_sub_f32 on the same registers.
# usage (GPU box): bash scripts/k3_race/replay_ab.sh [ITERS] -> gpurun_out/r03/replay_ab.log
IT=${1:-6000}
O=gpurun_out/r03; mkdir -p $O
LOG=$O/replay_ab.log
echo "#### $(date -u) $(hostname)" > $LOG
for rnd in 1 2; do
for v in pre0 preA preB; do
  export ENF_HIP_LIB=$PWD/variants/libenf_$v.so
  echo "== $v (round $rnd)" | tee -a $LOG
  POLLUTE=1 timeout -k 10 600 python scripts/k3_race/store_probe.py $IT 128 1 bf16 2>&1 | grep -v "^it \|signature\|amdgpu.ids" | tail -1 | tee -a $LOG
  if [ "$v" = "pre0" ] && [ "$rnd" = "1" ] && grep -q "differ from the first 0," $LOG; then echo "control clean on this box: stop" | tee -a $LOG; exit 0; fi
done
done
