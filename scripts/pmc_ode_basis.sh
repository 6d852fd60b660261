#!/bin/bash
# SQ counter passes (one --pmc set per run) over the fused kernel-basis kernels of the latent ODE (scripts/probe_ode_basis.py);
# summary -> gpurun_out/pmc_<tag>/summary.txt.   usage: scripts/pmc_ode_basis.sh TAG  (on the GPU box through gpurun)
TAG=${1:-ode_basis}
REPO=$PWD
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o p -- python3 $REPO/scripts/probe_ode_basis.py > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; }
done
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$OUT/summary.txt","w") as o:
    for k,d in acc.items():
        if "ode_basis" not in k: continue
        o.write(k+"\n")
        for c,v in sorted(d.items()):
            o.write(f"  {c:32s} n={len(v):3d} mean={sum(v)/len(v):.4g}\n")
print(open("$OUT/summary.txt").read())
PY
