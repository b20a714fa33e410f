"""Build libsdeng.so (gfx950) in-tree with hipcc.

    python -m sde_sampler_lrds_amd.build [-j N] [--force]

The simulate kernel is a template over (feature tiles, reference kind, in-loop score kind, update form);
each instantiation is its own translation unit (generated under csrc/gen/) so they compile in parallel.
hipcc cross-compiles for gfx950 without a GPU.
"""
from __future__ import annotations

import argparse
import hashlib
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
GEN = os.path.join(CSRC, "gen")
# experiment knobs: SDENG_OUT = another library path (its objects go to csrc/obj_<name>/), SDENG_PACKED=0/1 below
LIB = os.environ.get("SDENG_OUT") or os.path.join(PKG, "libsdeng.so")
OBJ = os.path.join(CSRC, "obj" if not os.environ.get("SDENG_OUT") else "obj_" + os.path.basename(LIB).split(".")[0])
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: the integrator and log-weight updates must round like the reference's separate
# torch ops (no silent a*b+c fusion); fused multiply-adds are written explicitly where wanted.
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-fPIC", "-Wno-comment", "-Wno-unused-command-line-argument"]
# Packed fp32 (v_pk_{add,mul,fma}_f32) stays disabled.  With two waves per SIMD the build that uses them is not
# reproducible on MI355X (the pre-empted wave of a SIMD gets wrong values; DESIGN 4a, profiles/r02_packed_fp32_hazard_experiments.log)
# and it is slower anyway (6.11 vs 5.92 ms on cfg 2).  SDENG_PACKED=1 builds that variant for A/B runs only.
PACKED = os.environ.get("SDENG_PACKED", "0") == "1"
if not PACKED:
    FLAGS += ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
if os.environ.get("SDENG_DEFS"):  # experiment knob: extra -D flags (ablation builds)
    FLAGS += ["-D" + d for d in os.environ["SDENG_DEFS"].split()]
if os.environ.get("SDENG_CXXFLAGS"):  # experiment knob: extra compiler flags (e.g. "-mllvm -amdgpu-sched-strategy=iterative-ilp")
    FLAGS += os.environ["SDENG_CXXFLAGS"].split()
if os.environ.get("SDENG_WAVES"):  # experiment knob: waves per workgroup (8 = 2 per SIMD, 4 = 1 per SIMD)
    FLAGS.append("-DSD_WAVES=" + os.environ["SDENG_WAVES"])

DTS, REFS, SCS, FORMS = (1, 2, 3, 4, 5, 6, 7, 8), (0, 1, 2, 3), (0, 1, 2), (0, 1)  # DTS: feature tiles of 16, NT = ceil(d / 16)
DTS_FULL = (1, 2, 3, 4, 6, 8)  # full-covariance reference: two staged pieces of NT/2 output tiles when NT > 4


def sources():
    os.makedirs(GEN, exist_ok=True)
    srcs = [os.path.join(CSRC, "sdeng_api.hip"), os.path.join(CSRC, "prep_kernels.hip"), os.path.join(CSRC, "cmcd_inst.hip"),
            os.path.join(CSRC, "euler_inst.hip")]
    for dt in DTS:  # fused forward + backward of the drift net (grad_kernel.hpp): the training direction's batched control pass
        path = os.path.join(GEN, f"vjp_{dt}.hip")
        _write_if_changed(path, '#include "../grad_kernel.hpp"\n' + f"SD_DEFINE_VJP({dt})\n")
        srcs.append(path)
    for dt in DTS:  # CMCD kernels (3 target kinds each)
        path = os.path.join(GEN, f"cmcd_{dt}.hip")
        _write_if_changed(path, '#include "../cmcd_kernel.hpp"\n'
                                f"int sd_launch_cmcd_{dt}(const CmcdArgs& a, int grid, hipStream_t s) {{ return launch_cmcd<{dt}>(a, grid, s); }}\n")
        srcs.append(path)
    for dt in (1, 2, 3, 4):  # in-loop logistic-regression score (SC = 3): no reference, d <= 64, forward forms only
        path = os.path.join(GEN, f"sim_{dt}_0_3.hip")
        _write_if_changed(path, '#include "../sim_kernel.hpp"\n' + "".join(f"SD_DEFINE_SIM({dt}, 0, 3, {fm})\n" for fm in FORMS))
        srcs.append(path)
    for dt in DTS_FULL:  # full-covariance mixture reference (REF = 4): ClippedCtrl; forward forms and the noising loop (3 = EUBO)
        path = os.path.join(GEN, f"sim_{dt}_4_0.hip")
        _write_if_changed(path, '#include "../sim_kernel.hpp"\n' + "".join(f"SD_DEFINE_SIM({dt}, 4, 0, {fm})\n" for fm in FORMS + (3,)))
        srcs.append(path)
    for dt in DTS_FULL:  # full-covariance mixture TARGET of a score control, held in the reference slot (REF = 4, SC = 4): forward forms
        path = os.path.join(GEN, f"sim_{dt}_4_4.hip")
        _write_if_changed(path, '#include "../sim_kernel.hpp"\n' + "".join(f"SD_DEFINE_SIM({dt}, 4, 4, {fm})\n" for fm in FORMS))
        srcs.append(path)
    for dt in (5, 6, 7, 8):  # low-latency small-batch kernels (split_kernel.hpp): a tile's features over four waves, d > 64
        path = os.path.join(GEN, f"split_{dt}.hip")
        _write_if_changed(path, '#include "../split_kernel.hpp"\n' + f"SD_DEFINE_SPLIT({dt})\n")
        srcs.append(path)
    for dt in DTS:  # shared-variance mixture reference on the matrix pipe (REF = 5): ClippedCtrl, forward forms
        path = os.path.join(GEN, f"sim_{dt}_5_0.hip")
        _write_if_changed(path, '#include "../sim_kernel.hpp"\n' + "".join(f"SD_DEFINE_SIM({dt}, 5, 0, {fm})\n" for fm in FORMS))
        srcs.append(path)
    for dt in DTS:
        for rf in REFS:
            for sc in SCS:
                path = os.path.join(GEN, f"sim_{dt}_{rf}_{sc}.hip")
                # 3 = SDENG_FORM_EUBO: reference-SDE losses with a ClippedCtrl, or DIS (score control, no reference)
                forms = FORMS + ((3,) if ((sc == 0) != (rf == 0)) else ())
                body = '#include "../sim_kernel.hpp"\n' + "".join(f"SD_DEFINE_SIM({dt}, {rf}, {sc}, {fm})\n" for fm in forms)
                _write_if_changed(path, body)
                srcs.append(path)
        path = os.path.join(GEN, f"ctrl_{dt}.hip")
        _write_if_changed(path, '#include "../sim_kernel.hpp"\n' + "".join(f"SD_DEFINE_CTRL({dt}, {sc})\n" for sc in SCS))
        srcs.append(path)
    return srcs


def _resource_remarks(stderr):
    """-Rpass-analysis=kernel-resource-usage remarks -> [{name, VGPRs, ScratchSize..., Occupancy...}]"""
    import re
    out, cur = [], None
    for line in stderr.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            out.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    return out


def _write_if_changed(path, body):
    if os.path.exists(path) and open(path).read() == body:
        return
    with open(path, "w") as f:
        f.write(body)


def _digest(with_sources=True):
    """Hash of the flags and the headers (+ the .hip sources): the library stamp, or the part every object depends on."""
    h = hashlib.sha256()
    for root in (CSRC, os.path.join(PKG, "..", "include")):
        for name in sorted(os.listdir(root)):
            if name.endswith((".hpp", ".h") + ((".hip",) if with_sources else ())):
                h.update(open(os.path.join(root, name), "rb").read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(jobs: int | None = None, force: bool = False, verbose: bool = True) -> str:
    stamp = os.path.join(OBJ, "digest.txt")
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == dig:
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    srcs = sources()
    jobs = jobs or min(8, os.cpu_count() or 1)

    usage = {}
    common = _digest(with_sources=False)

    def compile_one(src):
        """One translation unit -> object; skipped when neither the flags, nor a header, nor this source changed."""
        import json
        obj = os.path.join(OBJ, os.path.basename(src).replace(".hip", ".o"))
        key = hashlib.sha256((common + open(src).read()).encode()).hexdigest()
        meta = obj + ".json"
        if not force and os.path.exists(obj) and os.path.exists(meta):
            m = json.load(open(meta))
            if m.get("key") == key:
                usage[os.path.basename(src)] = m["usage"]
                return obj
        cmd = [HIPCC, *FLAGS, "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
        usage[os.path.basename(src)] = _resource_remarks(r.stderr)
        json.dump({"key": key, "usage": usage[os.path.basename(src)]}, open(meta, "w"))
        return obj

    with ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(compile_one, srcs))
    with open(os.path.join(OBJ, "kernel_resources.txt"), "w") as f:  # registers / scratch / occupancy of every kernel (compiler remarks)
        f.write("# unit kernel VGPRs AGPRs scratch_bytes_per_lane occupancy_waves_per_SIMD LDS_bytes\n")
        for unit in sorted(usage):
            for k in usage[unit]:
                f.write(f"{unit} {k['name']} {k.get('VGPRs', '?')} {k.get('AGPRs', '?')} {k.get('ScratchSize [bytes/lane]', '?')} "
                        f"{k.get('Occupancy [waves/SIMD]', '?')} {k.get('LDS Size [bytes/block]', '?')}\n")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    with open(stamp, "w") as f:
        f.write(dig)
    if verbose:
        print(f"built {LIB} ({os.path.getsize(LIB) / 1e6:.1f} MB, {len(objs)} objects)")
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("-j", type=int, default=None)
    ap.add_argument("--force", action="store_true")
    a = ap.parse_args()
    build(a.j, a.force)
