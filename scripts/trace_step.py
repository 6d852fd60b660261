"""Timeline of one fit+decode step of bench.py from a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d OUT -o b -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline
    python scripts/trace_step.py OUT/b_kernel_trace.csv
Prints every launch of the 4th timed step (start offset, duration, gap to the previous kernel's end) and the totals."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
k = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
# a step ends with the decode-shape forward pair kernel (z-fold instantiation <.., true, true>) and its tail
dec = [i for i, (n, s, e) in enumerate(k) if re.search(r"enf_pair_fwd_kernel<\d+, \d+, (true|false), true(, [^>]*)?>", n)]
a, b = dec[3], dec[4]
seg = k[a + 2:b + 2]
t0, busy, prev = seg[0][1], 0, None
for n, s, e in seg:
    gap = 0 if prev is None else s - prev
    nm = re.sub(r"void |at::native::|\(anonymous namespace\)::", "", n)[:80]
    print(f"{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:8.1f} us  gap {gap / 1e3:7.1f}  {nm}")
    busy += e - s
    prev = max(prev or 0, e)
print(f"launches {len(seg)}  busy {busy / 1e6:.3f} ms  span {(seg[-1][2] - t0) / 1e6:.3f} ms")
