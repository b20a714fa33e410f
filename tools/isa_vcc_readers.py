"""v_cndmask_b32 in its VOP2 form (mask in vcc) costs ~4 plain vector instructions on gfx950 unless it directly follows the v_cmp that wrote
vcc (tools/ubench/cndmask_cost.hip).  Per kernel: how many e32 cndmasks are the 1st / 2nd / later reader of one vcc value, and how far from its writer.
usage: python tools/isa_vcc_readers.py file.s [name-substring] [-v]"""
import collections
import re
import sys

s = open(sys.argv[1]).read().splitlines()
want = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else ""
verbose = "-v" in sys.argv
starts = [(i, m.group(1)) for i, l in enumerate(s) for m in [re.match(r"(_Z\S+):\s", l + " ")] if m and not l.startswith(".")]
ends = [i for i, l in enumerate(s) if l.strip().startswith("s_endpgm")]
for i, name in starts:
    if want not in name:
        continue
    e = min((x for x in ends if x > i), default=len(s))
    f = [(k, l.strip()) for k, l in enumerate(s[i:e]) if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    mf = [n for n, (_, l) in enumerate(f) if "v_mfma" in l]
    if not mf:
        continue
    order = collections.Counter()
    dist = collections.Counter()
    writer, nth = None, 0
    for n in range(mf[0], mf[-1]):
        k, l = f[n]
        op, _, ops = l.partition(" ")
        if op.startswith("v_cndmask_b32_e32") or (op.startswith(("v_addc", "v_subb")) and "vcc" in ops.split(",")[-1]):
            nth += 1
            order[min(nth, 3)] += 1
            dist[min(n - writer, 8) if writer is not None else 99] += 1
            if verbose and (nth > 1 or (writer is not None and n - writer > 2)):
                print(f"   line {k}: reader #{nth}, {n - writer if writer is not None else -1} after the writer: {l}")
        if re.match(r"v_cmpx?_\w+_e32", op) or ops.lstrip().startswith("vcc") or re.search(r"^v\d+, vcc,", ops):
            writer, nth = n, 0
    print(name, "| between first and last MFMA: e32 cndmask/addc readers by order of reading", dict(order), "by distance to the writer", dict(sorted(dist.items())))
