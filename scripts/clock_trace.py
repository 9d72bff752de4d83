"""Step time per group of 10 steps over a long run, with the SMI clock/power readings beside it."""
import os, sys, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pai_bootstrap
pai = pai_bootstrap.load()
dev = torch.device("cuda", 0)
torch.manual_seed(0)
m = pai.Pix2Pix(1, 1, (1, 2, 4, 8, 8, 8, 8, 8), 0.0, "gan").to(dev)
m.set_precision("bf16-mixed"); m.train(); m.optimizers()
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.random((64, 1, 256, 256), dtype=np.float32) * 2 - 1).to(dev)
t = torch.from_numpy(rng.random((64, 1, 256, 256), dtype=np.float32) * 2 - 1).to(dev)
def smi():
    try:
        o = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True, timeout=20).stdout
        keep = [l.split(":", 1)[1].strip() for l in o.splitlines() if ("sclk" in l or "Power" in l or "junction" in l.lower() or "mclk" in l)]
        return " | ".join(keep)
    except Exception as e:
        return repr(e)
t00 = time.perf_counter()
n_groups = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for gidx in range(n_groups):
    torch.cuda.synchronize(); tg = time.perf_counter()
    for i in range(10):
        m.training_step((x, t), i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - tg) / 10 * 1e3
    extra = smi() if gidx % 5 == 4 else ""
    print(f"t={time.perf_counter()-t00:6.2f}s group {gidx:3d}: {dt:7.3f} ms/step {extra}", flush=True)
