"""Interleaved A/B/... of bench.py under different environments on one box (the chip drifts a few per cent
with temperature, so variants are alternated and the median per variant is reported).
usage: python scripts/abn.py [-r ROUNDS] "name:ENV=1 ENV2=x" "name2:" ...   (extra bench args after --)"""
import json, os, statistics, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
rounds = 3
if args and args[0] == "-r":
    rounds = int(args[1]); args = args[2:]
extra = []
if "--" in args:
    i = args.index("--"); extra = args[i + 1:]; args = args[:i]
variants = []
for a in args:
    name, _, envs = a.partition(":")
    env = dict(os.environ)
    for kv in envs.split():
        k, _, v = kv.partition("=")
        env[k] = v
    variants.append((name, env))
res = {n: [] for n, _ in variants}
for r in range(rounds):
    for name, env in variants:
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--no-kernel-events"] + extra,
                             env=env, capture_output=True, text=True, timeout=400).stdout.strip().splitlines()
        try:
            res[name].append(json.loads(out[-1])["ms_per_step"])
        except Exception:
            res[name].append(float("nan"))
for name, _ in variants:
    v = res[name]
    print(f"{name:24s} median {statistics.median(v):7.3f} ms  runs {' '.join(f'{x:.3f}' for x in v)}")
