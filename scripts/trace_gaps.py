"""Idle-gap analysis of a rocprofv3 --kernel-trace CSV of bench.py: wall / busy / per-queue time per step over the last
steps of the run (steps are delimited by the generator's Adam launch)."""
import collections
import csv
import glob
import sys


def main(d, nsteps=5, marker="adam_k"):
    f = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    marker_name = marker.split(":")[0]
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in rows)
    adam = [i for i, e in enumerate(ev) if marker_name in e[2]]
    per_step = 2 if marker == "adam_k" else int(marker.split(":")[1]) if ":" in marker else 1
    i0, i1 = adam[-per_step * nsteps - 1], adam[-1]
    seg = ev[i0 + 1:i1 + 1]
    t0, t1 = seg[0][0], seg[-1][1]
    iv = sorted((s, e) for s, e, _, _ in seg)
    busy, gaps = 0, []
    cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce:
            busy += ce - cs
            gaps.append(s - ce)
            cs, ce = s, e
        else:
            ce = max(ce, e)
    busy += ce - cs
    byq = collections.Counter()
    byk = collections.Counter()
    for s, e, n, q in seg:
        byq[q] += e - s
        byk[n.split("(")[0][:60]] += e - s
    print(f"wall {1e-6 * (t1 - t0) / nsteps:.3f} ms/step  busy(union) {1e-6 * busy / nsteps:.3f}  "
          f"sum {1e-6 * sum(e - s for s, e, _, _ in seg) / nsteps:.3f}  launches/step {len(seg) / nsteps:.0f}")
    print("per queue:", {k: round(1e-6 * v / nsteps, 3) for k, v in byq.items()})
    print(f"gaps: {len(gaps) / nsteps:.0f}/step, total {1e-6 * sum(gaps) / nsteps:.3f} ms/step, largest (us):",
          [round(g / 1e3, 1) for g in sorted(gaps, reverse=True)[:12]])
    for k, v in byk.most_common(14):
        print(f"  {1e-6 * v / nsteps:7.3f} ms/step  {k}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 5, sys.argv[3] if len(sys.argv) > 3 else "adam_k")
