"""Assembly-level patches for hazard experiments (see asm_variant.sh).

    patch_asm.py in.s out.s "<spec>[;<spec>...]"

spec = <where>:<what>[:<kernel substring>]
  where   after_mfma | before_mfma | before_pk | after_pk | before_pk_reading_mfma (a v_pk_* whose sources were written by an
          MFMA at most 24 wait states earlier) | after_mfma_c_dead (an MFMA whose SrcC registers are not its destination) | none
  what    nopN (s_nop N-1, i.e. N wait states)
Only kernels whose mangled name contains the substring are patched (default: every kernel).
"""
import re
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from isa_hazard_scan import parse, regs  # noqa: E402

src, dst, specs = sys.argv[1], sys.argv[2], sys.argv[3]
lines = open(src).read().split("\n")
out = []
counts = {}
for spec in [s for s in specs.split(";") if s and s != "none"]:
    parts = spec.split(":")
    where, what = parts[0], parts[1]
    ksub = parts[2] if len(parts) > 2 else ""
    n = int(what[3:])
    nop = [f"\ts_nop {min(7, n - 1 - 8 * i)}" for i in range((n + 7) // 8) if n - 8 * i > 0]
    res = []
    kernel = None
    hist = []
    pos = 0
    for raw in lines:
        m = re.match(r"^(_Z\w+):", raw)
        if m:
            kernel, hist, pos = m.group(1), [], 0
        p = parse(raw)
        active = kernel is not None and ksub in kernel
        pre = post = False
        if p is not None and active:
            op, ops = p
            if op.startswith("v_mfma"):
                d, a, b, c = (regs(o) for o in ops[:4])
                pre = where == "before_mfma"
                post = where == "after_mfma" or (where == "after_mfma_c_dead" and c and not c <= d)
            elif op.startswith("v_pk_"):
                pre = where == "before_pk"
                post = where == "after_pk"
                if where == "before_pk_reading_mfma":
                    u = set().union(*[regs(o) for o in ops[1:]])
                    pre = any(pos - hp <= 24 and (u & hd) for hp, hd in hist)
            if op.startswith("v_mfma"):
                hist.append((pos, regs(ops[0])))
                hist = hist[-32:]
            pos += (int(ops[0], 0) + 1) if op == "s_nop" else 1
        if pre:
            res += nop
            counts[spec] = counts.get(spec, 0) + 1
        res.append(raw)
        if post:
            res += nop
            counts[spec] = counts.get(spec, 0) + 1
        if p is not None and p[0] == "s_endpgm":
            kernel = None
    lines = res
open(dst, "w").write("\n".join(lines))
print("patched:", counts)
