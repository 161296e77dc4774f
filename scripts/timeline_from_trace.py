#!/usr/bin/env python3
"""Where a pipelined frame's time goes, from a rocprofv3 --kernel-trace CSV of `python bench.py` (no --stats needed):
  cd /tmp && rocprofv3 --kernel-trace -d out -o t --output-format csv -- python bench.py --no-cpu-baseline --steps 30 --warmup 12
  python scripts/timeline_from_trace.py out/.../t_kernel_trace.csv [frames]
Prints, averaged over the last `frames` frames (a frame = one k_rtdgi_trace_fused launch to the next):
  * per kernel: launches per frame, mean duration, mean start offset inside the frame, the queue it ran on;
  * per queue: busy time per frame; and the time during which 1 / 2 / 3+ kernels were running at once.
"""
import csv, sys, collections

path = sys.argv[1]
n_frames = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Queue_Id"]))
rows.sort()
marks = [s for s, e, n, q in rows if n.startswith("k_rtdgi_trace_fused")]
if len(marks) < n_frames + 2:
    sys.exit("not enough frames in the trace")
t0, t1 = marks[-n_frames - 1], marks[-1]
frame_ns = (t1 - t0) / n_frames
sel = [(s, e, n, q) for s, e, n, q in rows if t0 <= s < t1]
per = collections.defaultdict(lambda: [0, 0.0, 0.0, set()])
for s, e, n, q in sel:
    # offset inside the frame the launch belongs to
    k = max(i for i, m in enumerate(marks) if m <= s)
    p = per[n]
    p[0] += 1; p[1] += e - s; p[2] += s - marks[k]; p[3].add(q)
print(f"frame {frame_ns / 1e3:.1f} us over {n_frames} frames")
print(f"{'kernel':58s} {'n/frame':>7s} {'mean us':>8s} {'us/frame':>9s} {'start us':>9s}  queue")
for n, (c, dur, off, qs) in sorted(per.items(), key=lambda kv: kv[1][2] / kv[1][0]):
    print(f"{n[:58]:58s} {c / n_frames:7.2f} {dur / c / 1e3:8.1f} {dur / n_frames / 1e3:9.1f} {off / c / 1e3:9.1f}  {','.join(sorted(qs))}")
busy = collections.defaultdict(float)
for s, e, n, q in sel:
    busy[q] += e - s
for q, b in sorted(busy.items()):
    print(f"queue {q}: busy {b / n_frames / 1e3:.1f} us / frame ({100 * b / n_frames / frame_ns:.0f} %)")
ev = []
for s, e, n, q in sel:
    ev.append((s, 1)); ev.append((min(e, t1), -1))
ev.sort()
depth, last, conc = 0, t0, collections.defaultdict(float)
for t, d in ev:
    conc[min(depth, 3)] += t - last
    last = t; depth += d
conc[min(depth, 3)] += t1 - last
print("kernels in flight: " + ", ".join(f"{k}{'+' if k == 3 else ''}: {v / n_frames / 1e3:.1f} us" for k, v in sorted(conc.items())))
