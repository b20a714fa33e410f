"""Opcode census of the step loop of one k_simulate instantiation.

    hipcc ... -save-temps=obj -c gen/sim_8_2_0.hip -o /tmp/s.o
    python tools/isa_census.py /tmp/sim_8_2_0-hip-amdgcn-amd-amdhsa-gfx950.s _Z10k_simulateILi8ELi2ELi0ELi0ELi0EEv7SimArgs

Prints the instruction mix of the depth-2 loop that contains the MFMAs (the per-step loop) with an issue-cycle
estimate (4 cycles per full-rate VALU op, 16 for quarter-rate: transcendentals, 32-bit integer multiplies;
MFMA 16x16x32 f16 = 16).
"""
import collections
import re
import sys

S, KN = sys.argv[1], sys.argv[2]
L = open(S).read().split("\n")
start = [i for i, l in enumerate(L) if l.startswith(KN + ":")][0]
end = [i for i, l in enumerate(L) if i > start and l.strip().startswith("s_endpgm")][0]
body = L[start:end]
best = None
for i, l in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if not m or i + 1 >= len(body) or "Depth=2" not in (body[i] + body[i + 1] + (body[i + 2] if i + 2 < len(body) else "")):
        continue
    lab = m.group(1)
    br = [j for j, x in enumerate(body) if j > i and ("s_cbranch" in x or "s_branch" in x) and x.split()[-1] == lab]
    if br and any("v_mfma" in x for x in body[i:br[-1] + 1]):
        best = body[i:br[-1] + 1]
        break
if best is None:
    sys.exit("step loop not found")
QUARTER = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32",
           "v_mul_hi_i32", "v_mad_i64_i32")
c = collections.Counter()
cyc = collections.Counter()
for l in best:
    l = l.strip()
    if not l or l.startswith(";") or l.startswith("."):
        continue
    op = l.split()[0]
    c[op] += 1
    if op.startswith("v_mfma"):
        cyc["mfma"] += 16
    elif op.startswith(QUARTER):
        cyc["valu_quarter"] += 16
    elif op.startswith("v_"):
        cyc["valu"] += 4
    elif op.startswith("ds_"):
        cyc["lds"] += 4
    elif op.startswith(("s_",)):
        cyc["salu"] += 4
    else:
        cyc["mem"] += 4
tot = sum(c.values())
print("instructions in step loop:", tot)
print("issue-cycle estimate:", dict(cyc), "sum", sum(cyc.values()))
for k, v in c.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 60):
    print(f"{v:6d} {k}")
