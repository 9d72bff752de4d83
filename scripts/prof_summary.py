"""Summarise a rocprofv3 --kernel-trace --stats csv directory: per-kernel totals per step and the
per-launch timeline of the last step."""
import csv, glob, sys
d = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 0
stats = glob.glob(d + "/*/*_kernel_stats.csv")[0]
trace = glob.glob(d + "/*/*_kernel_trace.csv")[0]
rows = list(csv.DictReader(open(stats)))
if not steps:   # one ssim_k launch per training step (clock-ramp and warm-up steps included)
    steps = max(int(r['Calls']) for r in rows if 'ssim_k' in r['Name'])
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"kernel time per step: {tot/1e6/steps:.2f} ms")
for r in rows[:30]:
    print(f"{float(r['TotalDurationNs'])/1e6/steps:7.3f} ms/step {float(r['Percentage']):5.1f}% calls/step {int(r['Calls'])/steps:5.1f} avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:80]}")
if "--timeline" in sys.argv:
    rows = list(csv.DictReader(open(trace)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('void ssim_k')]
    step = rows[idx[-2]:idx[-1]]
    print(len(step), "launches in the last step")
    for r in step:
        dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        if dur > float(sys.argv[sys.argv.index("--timeline") + 1]):
            print(f"{dur:8.1f} us blocks {int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']):>7} vgpr {r['VGPR_Count']:>4} {r['Kernel_Name'][:70]}")
