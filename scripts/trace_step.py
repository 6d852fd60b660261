"""Timeline of the bench's fit+decode steps from a rocprofv3 kernel trace: python scripts/trace_step.py <kernel_trace.csv>
Prints, for the last full step, every launch (start offset, duration, gap to the previous kernel's end) and the totals."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
k = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
# a step ends with the decode-shape forward pair kernel (z-fold): find the last two of them
idx = [i for i, (n, s, e) in enumerate(k) if "enf_pair_fwd_kernel" in n and "true, true" in n.replace(" ", "").replace(",", ", ")]
if len(idx) < 3:
    idx = [i for i, (n, s, e) in enumerate(k) if "enf_pair_fwd_kernel" in n and (e - s) > 900000]
a, b = idx[-3], idx[-2]
# step = from after the tail that follows decode a, to the end of the tail following decode b
seg = k[a + 2:b + 2]
t0 = seg[0][1]
busy, prev = 0, None
for n, s, e in seg:
    gap = 0 if prev is None else s - prev
    print(f"{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:8.1f} us  gap {gap / 1e3:7.1f}  {n[:70]}")
    busy += e - s
    prev = max(prev or 0, e)
print(f"launches {len(seg)}  busy {busy / 1e6:.3f} ms  span {(seg[-1][2] - t0) / 1e6:.3f} ms")
