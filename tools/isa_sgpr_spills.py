"""SGPR spills (v_readlane / v_writelane) inside the step loop of one kernel of a hipcc -S file, by the instruction that consumes them.
usage: python tools/isa_sgpr_spills.py file.s <mangled kernel name prefix>"""
import collections
import re
import sys

s = open(sys.argv[1]).read().splitlines()
st = [i for i, l in enumerate(s) if l.startswith(sys.argv[2])][0]
en = min([i for i, l in enumerate(s) if i > st and l.strip().startswith("s_endpgm")] + [len(s)])
f = s[st:en]
is_ins = lambda x: x.startswith("\t") and not x.strip().startswith((".", ";"))
lab = {m.group(1): k for k, l in enumerate(f) for m in [re.match(r"(\.LBB\d+_\d+):", l)] if m}
loops = []
for k, l in enumerate(f):
    m = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
    if m and m.group(1) in lab and lab[m.group(1)] < k:
        a = lab[m.group(1)]
        n = sum(1 for x in f[a:k] if is_ins(x))
        if n > 2500:
            loops.append((a, k, n))
a, b, n = loops[0]
c = collections.Counter(x.split()[0] for x in f[a:b] if is_ins(x))
print(f"step loop: lines {a}-{b}, {n} instructions; v_readlane {c['v_readlane_b32']}, v_writelane {c['v_writelane_b32']}, s_nop {c['s_nop']}, "
      f"valu {sum(v for k, v in c.items() if k.startswith('v_') and 'mfma' not in k)}, mfma {sum(v for k, v in c.items() if 'mfma' in k)}")
uses = collections.Counter()
for i in range(a, b):
    m = re.match(r"\s+v_readlane_b32 (s\d+), (v\d+), (\d+)", f[i])
    if not m:
        continue
    sr = int(m.group(1)[1:])
    for j in range(i + 1, min(i + 60, b)):
        if not is_ins(f[j]) or "readlane" in f[j]:
            continue
        hit = re.search(r"\bs%d\b" % sr, f[j]) or any(int(x) <= sr <= int(y) for x, y in re.findall(r"s\[(\d+):(\d+)\]", f[j]))
        if hit:
            uses[f[j].split()[0]] += 1
            break
print(uses.most_common(12))
