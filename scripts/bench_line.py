import sys, json
tag = sys.argv[1] if len(sys.argv) > 1 else ""
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d.get("roofline") or {}
print(tag, d["value"], d["ms_per_step"], r.get("achieved"), r.get("family_ms_per_step"))
