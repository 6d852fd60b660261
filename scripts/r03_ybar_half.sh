#!/bin/bash
# ENF_STAGE_YBAR_HALF: parity (forward / golden / configs / layers), HBM counters of a decode's pair kernel + tail with the bf16 and the fp32
# hand-off (FETCH_SIZE and WRITE_SIZE each in a pass of its own), then the bench against the previous commit
O=gpurun_out/r03
mkdir -p $O
R=$PWD
timeout -k 10 900 python -m pytest tests/test_gpu_forward.py tests/test_gpu_golden.py tests/test_gpu_configs.py tests/test_gpu_layers.py tests/test_gpu_reentrancy.py -m gpu -x -q > $O/yh_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/yh_tests.log
[ $rc = 0 ] || exit 1
cd /tmp && export TMPDIR=/tmp
for mode in half full; do for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/$O/yh_${mode}_$c
  if [ $mode = full ]; then export YBAR_FULL=1; else unset YBAR_FULL; fi
  timeout -k 10 240 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$O/yh_${mode}_$c -o p -- python3 $R/scripts/prof_decode.py > $R/$O/yh_${mode}_$c.log 2>&1 || { echo "pmc $mode $c failed"; tail -3 $R/$O/yh_${mode}_$c.log; }
done; done
unset YBAR_FULL
cd $R
python3 - <<'PY'
import csv, glob, collections
for mode in ("half", "full"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(f"gpurun_out/r03/yh_{mode}_{c}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0]
                if "pair_fwd" in k or "tail_fwd" in k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        fe, wr = (sum(d[c]) / max(len(d[c]), 1) for c in ("FETCH_SIZE", "WRITE_SIZE"))
        print(mode, k[-60:], "FETCH_KB %.0f WRITE_KB %.0f -> hbm MB %.1f" % (fe, wr, (2 * fe + wr) * 1024 / 1e6))
PY
for v in base default base default base default; do
  L=variants/libenf_$v.so; [ $v = default ] && L=
  ENF_HIP_LIB=$L timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-meta --no-ode --events-steps 0 --no-roofline > $O/yh_$v.json 2>$O/yh_$v.err || { echo "bench $v failed"; tail -5 $O/yh_$v.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('$O/yh_$v.json').readline()); print('$v', d['ms_per_step'], '%.2e' % d['accuracy']['mse_vs_oracle'])"
done
