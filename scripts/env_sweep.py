"""Re-check the launch-configuration defaults of the tile kernels on the bench: one bench.py run per environment."""
import json, os, subprocess, sys
SWEEP = [{}, {"PAI_FWD_MODE": "1"}, {"PAI_FWD_MODE": "2"}, {"PAI_PATCH_DBB": "7"}, {"PAI_PATCH_DBB": "1"}, {"PAI_PATCH_DBB": "2"},
         {"PAI_PATCH_DBB": "0"}, {"PAI_WGRAD_TARGET": "512"}, {"PAI_WGRAD_TARGET": "1024"}, {"PAI_WGRAD_TARGET": "1536"},
         {"PAI_NO_WPATCH": "1"}, {"PAI_WGRAD_MINROWS": "1024"}, {"PAI_TW_BLOCKS": "2048"}, {}]
for env in SWEEP:
    r = subprocess.run([sys.executable, "bench.py", "--steps", "30", "--warmup", "5", "--no-cpu-baseline", "--no-kernel-events"],
                       env=dict(os.environ, **env), capture_output=True, text=True)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
        print(f"{str(env):40s} {d['ms_per_step']:7.3f} ms  {d['value']:8.1f} img/s", flush=True)
    except Exception:
        print(env, "ERR", r.stderr[-300:], flush=True)
