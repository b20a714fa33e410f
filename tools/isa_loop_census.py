"""Per kernel of a hipcc -S file: instruction classes in the whole kernel and in its largest loop (the step loop).
usage: python tools/isa_loop_census.py file.s [name-substring]"""
import collections
import re
import sys

s = open(sys.argv[1]).read().splitlines()
want = sys.argv[2] if len(sys.argv) > 2 else ""
starts = [(i, m.group(1)) for i, l in enumerate(s) for m in [re.match(r"(_Z\S+):\s", l + " ")] if m and not l.startswith(".")]
ends = [i for i, l in enumerate(s) if l.strip().startswith("s_endpgm")]
for i, name in starts:
    if want not in name:
        continue
    e = min((x for x in ends if x > i), default=len(s))
    f = s[i:e]
    is_ins = lambda x: x.startswith("\t") and not x.strip().startswith((".", ";"))
    lab = {m.group(1): k for k, l in enumerate(f) for m in [re.match(r"(\.LBB\d+_\d+):", l)] if m}
    best = (0, 0, 0)
    for k, l in enumerate(f):
        m = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
        if m and m.group(1) in lab and lab[m.group(1)] < k:
            n = sum(1 for x in f[lab[m.group(1)]:k] if is_ins(x))
            if n > best[0]:
                best = (n, lab[m.group(1)], k)

    def census(lines):
        c = collections.Counter(x.split()[0] for x in lines if is_ins(x))
        g = lambda p: sum(v for k, v in c.items() if p(k))
        return dict(total=sum(c.values()), valu=g(lambda k: k.startswith("v_") and "mfma" not in k), mfma=g(lambda k: "mfma" in k),
                    nop=c["s_nop"], wait=c["s_waitcnt"], mov=g(lambda k: k.startswith("v_mov")), acc=g(lambda k: "accvgpr" in k),
                    scratch=g(lambda k: "scratch" in k), ds=g(lambda k: k.startswith("ds_")), vmem=g(lambda k: k.startswith(("global_", "buffer_"))),
                    salu=g(lambda k: k.startswith("s_")), branch=g(lambda k: k.startswith(("s_cbranch", "s_branch"))))
    print(name)
    print("   kernel:", census(f))
    loops = []
    for k, l in enumerate(f):
        m = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
        if m and m.group(1) in lab and lab[m.group(1)] < k:
            n = sum(1 for x in f[lab[m.group(1)]:k] if is_ins(x))
            if n > 1000:
                loops.append((lab[m.group(1)], k))
    for a, b in loops:  # the step loop is the innermost big one
        print("   loop  lines", a, "-", b, ":", census(f[a:b]))
