#!/usr/bin/env python3
"""Step timeline from a rocprofv3 --kernel-trace CSV: per hardware queue the busy time, the time no kernel runs on any
queue, and the kernels in flight beside the long ones -- what a per-kernel --stats table cannot show.
    python scripts/timeline.py <..._kernel_trace.csv> [steps_to_skip] [--seq]
--seq also lists the last step kernel by kernel: start offset, queue, duration, gap to the previous kernel of the same queue.
A "step" is delimited by the metrics_take_k launch of training_step."""
import collections
import csv
import sys


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0][:48]


def sequence(rows, ends):
    a, b = ends[-2], ends[-1]
    seg = rows[a + 1:b + 1]
    t0 = seg[0][0]
    last_end = {}
    qs = sorted({r[2] for r in seg})
    print(f"last step, {len(seg)} kernels, queues {qs}: start us | queue | duration us | gap on that queue us | kernel")
    for s, e, q, k in seg:
        gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
        print(f"  {(s - t0) / 1e3:8.1f} q{qs.index(q)} {(e - s) / 1e3:7.1f} {gap:7.1f}  {k}")
        last_end[q] = max(e, last_end.get(q, 0))


def main(path, skip=4, seq=False):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), short(r["Kernel_Name"])))
    rows.sort()
    # one metrics_take_k per training step (the streamed Adam launches are spread over the backward passes)
    ends = [i for i, r in enumerate(rows) if r[3].startswith("metrics_take")]
    if len(ends) < skip + 3:
        print("too few steps in the trace")
        return
    if seq:
        sequence(rows, ends)
    a, b = ends[skip], ends[-1]
    n = len(ends) - 1 - skip
    seg = rows[a + 1:b + 1]
    t0, t1 = seg[0][0], max(r[1] for r in seg)
    wall = (t1 - t0) / n
    per_q = collections.defaultdict(int)
    for s, e, q, _ in seg:
        per_q[q] += e - s
    # union of busy intervals
    ev = sorted([(s, 1) for s, e, q, _ in seg] + [(e, -1) for s, e, q, _ in seg])
    depth, last, idle, conc = 0, t0, 0, collections.defaultdict(int)
    for t, d in ev:
        if depth == 0:
            idle += t - last
        conc[min(depth, 3)] += t - last
        last = t
        depth += d
    print(f"{n} steps, {wall / 1e3:.1f} us per step between metrics_take_k launches")
    for q, v in sorted(per_q.items()):
        print(f"  queue {q}: kernels busy {v / n / 1e3:8.1f} us/step")
    print(f"  no kernel running: {idle / n / 1e3:.1f} us/step;  1 / 2 / 3+ kernels in flight: "
          f"{conc[1] / n / 1e3:.1f} / {conc[2] / n / 1e3:.1f} / {conc[3] / n / 1e3:.1f} us/step")
    # per kernel name: time while it is the ONLY kernel in flight vs overlapped
    solo = collections.defaultdict(int)
    tot = collections.defaultdict(int)
    active = {}
    ev2 = sorted([(s, 0, i) for i, (s, e, q, k) in enumerate(seg)] + [(e, -1, i) for i, (s, e, q, k) in enumerate(seg)])
    last = t0
    for t, kind, i in ev2:
        if len(active) == 1:
            solo[seg[next(iter(active))][3]] += t - last
        last = t
        if kind == 0:
            active[i] = 1
        else:
            active.pop(i, None)
    for s, e, q, k in seg:
        tot[k] += e - s
    print("  kernel: us/step total, of which alone on the GPU")
    for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:28]:
        print(f"    {k:48s} {v / n / 1e3:8.1f} {solo[k] / n / 1e3:8.1f}")


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "--seq"]
    main(args[0], int(args[1]) if len(args) > 1 else 4, "--seq" in sys.argv)
