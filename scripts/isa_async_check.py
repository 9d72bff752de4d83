"""Static check of the asynchronous-LDS-read discipline of the hand-scheduled K loops (gg_wgrad_patch3_k<.., PIPE = 1>).

Those loops issue their fragment reads (ds_read_b64_tr_b16) as inline asm and wait for them with explicit, partly COUNTED
s_waitcnt lgkmcnt(N); hipcc knows neither when a read delivers nor that a wait belongs to it.  Correctness rests on two
properties of the GENERATED code that the compiler does not guarantee (register allocation decides them):
  1. no instruction reads a register a ds_read writes before a wait has covered that read
     (DS operations complete in order: behind lgkmcnt(N) every read but the N youngest has delivered);
  2. no instruction WRITES such a register while the read is in flight either (the data would land on top of it).
This walks every innermost loop of a kernel that contains MFMAs (all paths of its body in program order; the loop is
walked twice so that reads crossing the back edge are seen) and reports violations.

  python scripts/isa_async_check.py file.s [kernel-name-substring ...]
Used by tests/test_isa_async.py (CPU: hipcc --save-temps cross-compiles without a GPU)."""
import re
import sys


def regs(tok):
    """VGPR numbers named by an operand token: v12, v[12:15]."""
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return [int(m.group(1))] if m else []


def operands(line):
    body = line.split(None, 1)[1] if " " in line else ""
    return [t.strip() for t in re.split(r",", body.split(" offset:")[0].split(" offen")[0]) if t.strip()]


def kernels(text):
    for m in re.finditer(r"^(_Z\S+):[^\n]*\n(.*?)\n\s*s_endpgm", text, re.S | re.M):
        yield m.group(1), [re.sub(r"\s*;.*$", "", l).strip() for l in m.group(2).split("\n")]


def loops_with_mfma(lines):
    labels = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r"^(\.LBB\S+):", l)] if m}
    out = []
    for i, l in enumerate(lines):
        m = re.search(r"\b(s_cbranch_\w+|s_branch)\s+(\.LBB\S+)", l)
        if m and labels.get(m.group(2), 1 << 30) < i:
            a = labels[m.group(2)]
            if any("v_mfma" in x for x in lines[a:i + 1]):
                out.append((a, i))
    # innermost only: drop loops that contain another one
    return [(a, b) for a, b in out if not any((c > a or d < b) and c >= a and d <= b for c, d in out if (c, d) != (a, b))]


def check_loop(lines, a, b):
    seq = 0                    # DS operations issued so far
    done = 0                   # DS operations known complete (sequence number)
    writer = {}                # vgpr -> sequence number of the ds_read that writes it
    bad = []
    body = [l for l in lines[a:b + 1] if l and not l.startswith((".", ";"))]
    for rnd in range(2):
        for l in body:
            op = l.split()[0]
            if op == "s_waitcnt":
                m = re.search(r"lgkmcnt\((\d+)\)", l)
                if m:
                    done = max(done, seq - int(m.group(1)))
                continue
            toks = operands(l)
            if op.startswith("ds_read"):
                seq += 1
                for r in regs(toks[0]):
                    if writer.get(r, 0) > done and rnd:
                        bad.append(("write into a register whose earlier read is in flight", l))
                    writer[r] = seq
                srcs = toks[1:]
            elif op.startswith(("ds_", "buffer_", "global_", "scratch_", "flat_")):
                if op.startswith("ds_"):
                    seq += 1
                srcs = toks
            else:
                srcs = toks[1:] if op.startswith("v_") or op.startswith("s_") else toks
                dst = toks[0] if toks else ""
                for r in regs(dst):
                    if writer.get(r, 0) > done and rnd:
                        bad.append(("write while a read into the register is in flight", l))
                if op.startswith("v_mfma") or op.startswith("v_pk_max") or op.startswith("v_dot2c"):
                    srcs = toks       # accumulate / in-place forms read their destination too
            for t in srcs:
                for r in regs(t):
                    if writer.get(r, 0) > done and rnd:
                        bad.append(("read of a register whose ds_read no wait has covered", l))
    return bad


def check_text(text, names=()):
    report = {}
    for name, lines in kernels(text):
        if names and not any(n in name for n in names):
            continue
        for a, b in loops_with_mfma(lines):
            bad = check_loop(lines, a, b)
            report[(name, a, b)] = bad
    return report


if __name__ == "__main__":
    rep = check_text(open(sys.argv[1]).read(), sys.argv[2:])
    rc = 0
    for (name, a, b), bad in rep.items():
        print(f"{name[:60]} loop lines {a}-{b}: {len(bad)} violations")
        for why, l in bad[:8]:
            print("   ", why, "|", l)
        rc |= bool(bad)
    sys.exit(rc)
