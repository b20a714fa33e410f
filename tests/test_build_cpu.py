"""Build-time guards of the HIP sources (CPU only: hipcc cross-compiles for gfx950 here).

DESIGN 4a: LLVM's hazard recognizer cannot see vector instructions written as inline asm, so an asm statement must never
contain one; and the emitted assembly of the headline unit is scanned for MFMA / asm adjacencies anyway."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sde_sampler_lrds_amd", "csrc")
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_no_vector_instruction_in_inline_asm():
    offenders = []
    for name in sorted(os.listdir(CSRC)):
        if not name.endswith((".hip", ".hpp")):
            continue
        text = open(os.path.join(CSRC, name)).read()
        for m in re.finditer(r"asm\s*(?:volatile)?\s*\(\s*((?:\"[^\"]*\"\s*)+)", text):
            body = "".join(re.findall(r"\"([^\"]*)\"", m.group(1)))
            if re.search(r"\b(v_|ds_|global_|buffer_|flat_)\w+", body):
                offenders.append((name, body.strip()[:60]))
    assert not offenders, f"vector/memory instructions inside asm statements are invisible to the hazard recognizer: {offenders}"


@pytest.mark.parametrize("unit", ["sim_8_2_0", "cmcd_4"])
def test_isa_hazard_scan_is_clean(unit, tmp_path):
    from isa_hazard_scan import scan

    from sde_sampler_lrds_amd import build

    build.sources()  # (re)generate csrc/gen
    src = os.path.join(build.GEN, unit + ".hip")
    out = str(tmp_path / (unit + ".s"))
    r = subprocess.run([build.HIPCC, *build.FLAGS, "--cuda-device-only", "-S", src, "-o", out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    text = open(out).read()
    assert "v_mfma_f32_16x16x32_f16" in text
    assert "v_pk_fma_f32" not in text and "v_pk_mul_f32" not in text and "v_pk_add_f32" not in text, "packed fp32 must stay disabled (DESIGN 4a)"
    assert scan(out, window=20, quiet=True) == []


def test_headline_kernels_keep_their_registers():
    """Registers / scratch / occupancy of the three bench kernels, from the compiler's own remarks of the build
    (csrc/obj/kernel_resources.txt): a spill or a lost wave in a step loop is a 5-15 % regression that no parity test sees.
    Every kernel built for three waves per SIMD (sim_kernel.hpp sd_waves_of) must fit that budget without scratch."""
    from sde_sampler_lrds_amd import build

    build.build(verbose=False)
    rows = {}
    for line in open(os.path.join(build.OBJ, "kernel_resources.txt")):
        if line.startswith("#"):
            continue
        unit, name, vgpr, agpr, scratch, occ, lds = line.split()
        rows[name] = (int(vgpr), int(scratch), int(occ))
    cfg2 = rows["_Z10k_simulateILi8ELi2ELi0ELi0ELi0EEv7SimArgs"]      # ManyModes d=128, mixture reference, EI
    cfg3 = rows["_Z10k_simulateILi7ELi0ELi2ELi1ELi0EEv7SimArgs"]      # PhiFour d=100, PIS, EM
    cfg4 = rows["_Z15k_simulate_cmcdILi4ELi0ELb0ELi0EEv8CmcdArgs"]    # logistic regression d=61, CMCD
    assert cfg2[1] == 0 and cfg2[2] >= 2, cfg2
    assert cfg3[1] <= 32 and cfg3[2] >= 3 and cfg3[0] <= 168, cfg3  # (any scratch must sit outside the step loop: ISA check below)
    assert cfg4[1] == 0 and cfg4[2] >= 2, cfg4
    three = [(n, r) for n, r in rows.items() if n.startswith("_Z10k_simulateI") and r[2] == 3]
    spilled = [(n, r) for n, r in three if r[1]]
    assert len(three) >= 30 and all(r[1] <= 32 for _, r in spilled), spilled[:5]
    # A few bytes of scratch are tolerated only OUTSIDE the step loop (a loop-invariant parked before it): checked in the ISA -- no
    # scratch instruction between the kernel's first and last matrix instruction.  (phi^4 at d = 128: 8 bytes since the range guard.)
    units = {line.split()[0] for line in open(os.path.join(build.OBJ, "kernel_resources.txt")) if not line.startswith("#") and line.split()[1] in dict(spilled)}
    for unit in sorted(units):
        out = os.path.join(build.OBJ, unit.replace(".hip", ".s"))
        r = subprocess.run([build.HIPCC, *build.FLAGS, "--cuda-device-only", "-S", os.path.join(build.GEN, unit), "-o", out], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        text = open(out).read()
        for name, _ in spilled:
            if name + ":" not in text:
                continue
            body = text[text.index(name + ":"):]
            body = body[:body.index(".Lfunc_end")].split("\n")
            mfma = [i for i, l in enumerate(body) if "v_mfma" in l]
            inside = [l.strip() for l in body[mfma[0]:mfma[-1]] if "scratch_" in l]
            assert not inside, (name, inside[:3])
