"""Hazards between inline-asm vector instructions and MFMAs that LLVM's hazard recognizer cannot see.

    hipcc <FLAGS> --cuda-device-only -S csrc/gen/sim_8_2_0.hip -o /tmp/s.s
    python tools/isa_hazard_scan.py /tmp/s.s [--window 20] [--quiet]

GCNHazardRecognizer inserts the wait states the gfx940/gfx950 rules ask for between an MFMA and a VALU instruction --
but only for instructions it recognises as VALU.  An `asm("v_cvt_pk_f16_f32 ...")` statement is an INLINEASM node:
its register defs and uses are allocated and scheduled like any other, yet none of the MAI hazard checks
(checkMAIHazards90A, checkMAIVALUHazards) run for it.  This script walks the emitted assembly (where the compiler brackets
such code with ;ASMSTART / ;ASMEND) and reports, per kernel, every place where an instruction inside an asm block

  WAR   writes a VGPR that an MFMA issued less than `window` wait states earlier reads as SrcC (and does not itself overwrite:
        vdst != src2, so the register is dead after the MFMA and the allocator hands it out again)
  WAW   writes a VGPR of that MFMA's destination
  RAW   reads a VGPR of that MFMA's destination
  DEF   defines a VGPR that an MFMA reads (A, B or C) fewer than 2 wait states later

Wait states are counted the way the hazard recognizer counts them: one per instruction, s_nop N = N + 1.
Exit status 1 if anything is found (build.py runs this as a build-time check on the shipped objects).
"""
import re
import sys


def regs(tok):
    """VGPR numbers named by one operand token: v12, v[4:7], -v3, |v3|."""
    tok = tok.strip().strip("-|").replace("neg(", "").replace("abs(", "").strip(")")
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def parse(line):
    line = line.split(";")[0].strip()
    if not line or line.startswith(".") or line.endswith(":"):
        return None
    parts = line.split(None, 1)
    op = parts[0]
    ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
    # operands like "v[0:3]" contain no comma; modifiers (op_sel:[0,0,1]) do -- drop everything from the first modifier on
    clean = []
    for o in ops:
        if ":" in o and not o.startswith(("v[", "s[", "a[", "-v[", "|v[")):
            break
        clean.append(o.split()[0] if o else o)
    return op, clean


def scan(path, window=20, quiet=False):
    lines = open(path).read().split("\n")
    findings = []
    kernel = None
    hist = []  # (wait-state position, op, operands, in_asm, line number)
    pos = 0
    in_asm = False
    for ln, raw in enumerate(lines, 1):
        s = raw.strip()
        m = re.match(r"^(_Z\w+):", raw)
        if m:
            kernel, hist, pos, in_asm = m.group(1), [], 0, False
            continue
        if "#ASMSTART" in s:
            in_asm = True
            continue
        if "#ASMEND" in s:
            in_asm = False
            continue
        p = parse(raw)
        if p is None or kernel is None:
            continue
        op, ops = p
        if op == "s_endpgm":
            kernel = None
            continue
        width = 1
        if op == "s_nop":
            width = int(ops[0], 0) + 1
        cur = (pos, op, ops, in_asm, ln)
        if op.startswith("v_mfma"):
            dst, a, b, c = (regs(o) for o in ops[:4])
            for (hp, hop, hops, hasm, hln) in reversed(hist):
                if pos - hp > 2:
                    break
                if hasm and hop.startswith("v_") and pos - hp < 2:
                    d = regs(hops[0]) if hops else set()
                    if d & (a | b | c):
                        findings.append((kernel, "DEF", hln, ln, pos - hp, f"{hop} {', '.join(hops)}  ->  {op} {', '.join(ops[:4])}"))
        elif in_asm and op.startswith("v_"):
            d = regs(ops[0]) if ops else set()
            u = set().union(*[regs(o) for o in ops[1:]]) if len(ops) > 1 else set()
            for (hp, hop, hops, hasm, hln) in reversed(hist):
                if pos - hp >= window:
                    break
                if not hop.startswith("v_mfma"):
                    continue
                mdst, ma, mb, mc = (regs(o) for o in hops[:4])
                ws = pos - hp - 1  # wait states between the two
                if d & mc and not (mc <= mdst):
                    findings.append((kernel, "WAR", hln, ln, ws, f"{hop} {', '.join(hops[:4])}  then  {op} {', '.join(ops)}"))
                if d & mdst:
                    findings.append((kernel, "WAW", hln, ln, ws, f"{hop} {', '.join(hops[:4])}  then  {op} {', '.join(ops)}"))
                if u & mdst:
                    findings.append((kernel, "RAW", hln, ln, ws, f"{hop} {', '.join(hops[:4])}  then  {op} {', '.join(ops)}"))
        hist.append(cur)
        if len(hist) > 4 * window:
            hist = hist[-2 * window:]
        pos += width
    if not quiet:
        by = {}
        for f in findings:
            by.setdefault((f[0], f[1]), []).append(f)
        for (k, kind), fs in sorted(by.items()):
            print(f"{k}: {kind} x{len(fs)}")
            for f in fs[:8]:
                print(f"    lines {f[2]}->{f[3]} ({f[4]} wait states between): {f[5]}")
        print(f"{path}: {len(findings)} unguarded asm/MFMA adjacencies")
    return findings


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    window = 20
    if "--window" in sys.argv:
        window = int(sys.argv[sys.argv.index("--window") + 1])
        args = [a for a in args if a != str(window)]
    bad = 0
    for path in args:
        bad += len(scan(path, window, "--quiet" in sys.argv))
    sys.exit(1 if bad else 0)
