"""Idle time between kernels in the replayed train step: reads a rocprofv3 kernel_trace.csv (start / end timestamps), takes the
window of the last N replayed steps (bench.py prints ms_per_step) and reports how much of it no kernel was running, and how the
gaps are distributed.  Usage (on the GPU box): python gap_analysis.py <kernel_trace.csv> <ms_per_step> [steps]"""
import csv, sys
f, ms = sys.argv[1], float(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", ""))))
rows.sort()
# the replayed steps are the densest stretch of the trace (the eager first step, the capture and the eager steps of the roofline
# record that follow the timed region launch the same kernels with host gaps between them): the window of `steps` step times
# that holds the most kernels
span_ns = int(ms * 1e6 * steps)
starts = [r[0] for r in rows]
best, t0 = -1, rows[0][0]
j = 0
for i in range(len(rows)):
    while rows[j][0] < rows[i][0] - span_ns:
        j += 1
    if i - j > best:
        best, t0 = i - j, rows[i][0] - span_ns
t1 = t0 + span_ns
win = [r for r in rows if r[0] >= t0 and r[1] <= t1]
busy, cur_s, cur_e, gaps = 0, None, None, []
for s, e, n, q in win:
    if cur_e is None:
        cur_s, cur_e = s, e
    elif s <= cur_e:
        cur_e = max(cur_e, e)
    else:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, n))
        cur_s, cur_e = s, e
busy += cur_e - cur_s
span = win[-1][1] - win[0][0]
ksum = sum(e - s for s, e, _, _ in win)
print(f"window {span/1e6:.2f} ms, {len(win)} kernels ({len(win)/steps:.0f} per step), kernel time summed {ksum/1e6:.2f} ms, "
      f"some kernel running {busy/1e6:.2f} ms, idle {(span-busy)/1e6:.2f} ms = {(span-busy)/span*100:.1f} % in {len(gaps)} gaps")
gs = sorted(g for g, _ in gaps)
if gs:
    import statistics
    print(f"gap median {statistics.median(gs)/1e3:.2f} us, mean {sum(gs)/len(gs)/1e3:.2f} us, p90 {gs[int(len(gs)*0.9)]/1e3:.2f} us, max {gs[-1]/1e3:.1f} us")
    big = sorted(gaps, reverse=True)[:12]
    for g, n in big:
        print(f"   {g/1e3:8.1f} us before {n[:90]}")
    buckets = [(0, 2), (2, 5), (5, 10), (10, 50), (50, 1e9)]
    for lo, hi in buckets:
        sel = [g for g in gs if lo * 1e3 <= g < hi * 1e3]
        print(f"   gaps {lo}-{hi if hi < 1e8 else 'inf'} us: {len(sel)/steps:.0f} per step, {sum(sel)/1e6/steps:.3f} ms per step")
# consistency of the window: launches per step of a kernel whose count is known (72 + 24 F(4x4,3x3) multiplies per fp32 step),
# kernel time per stream / queue, and the largest contributors
import collections
cnt = collections.Counter(n.split("(")[0][:60] for _, _, n, _ in win)
print("launches per step:", {k: round(v / steps, 1) for k, v in cnt.most_common(8)})
per_q = collections.defaultdict(int)
for s, e, n, q in win:
    per_q[q] += e - s
print("kernel time per queue (ms per step):", {k: round(v / 1e6 / steps, 2) for k, v in per_q.items()})
