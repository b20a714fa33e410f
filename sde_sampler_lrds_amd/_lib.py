"""ctypes binding of libsdeng.so (the C ABI of include/sdeng.h).

There is no CPU fallback: if the HIP library is missing or fails to load, importing this module's
``lib()`` raises.  ``python -m sde_sampler_lrds_amd.build`` (or ``__graft_entry__.build()``) builds it.
"""
from __future__ import annotations

import ctypes as C
import os

ABI_VERSION = 3
NCOEF = 16

FORM_LIN, FORM_EM, FORM_CMCD, FORM_EUBO, FORM_CMCD_EUBO = 0, 1, 2, 3, 4
FLAG_ITO, FLAG_INIT_LOGP, FLAG_TERM_REF, FLAG_TERM_TARGET, FLAG_SPLIT_TILES, FLAG_REMOVE_REF, FLAG_REUSE_PACK = 1, 2, 4, 8, 16, 32, 64
SPLIT_TILES_MAX_B = 8192  # SDENG_FLAG_SPLIT_TILES is honoured up to this batch size (sdeng.h, sdeng_api.hip split_eligible)
DIST_NONE, DIST_GMM_DIAG, DIST_GAUSS_DIAG, DIST_ISO_GAUSS, DIST_PHI4, DIST_LOGREG, DIST_GAUSS_FULL, DIST_RINGS, DIST_GMM_FULL = range(9)
CTRL_CLIPPED, CTRL_SCORE, CTRL_LERP, CTRL_NONE, CTRL_CANCEL_DRIFT = 0, 1, 2, 3, 4
REF_NONE, REF_GAUSS_DIAG, REF_GMM_DIAG, REF_GMM_FULL = 0, 1, 2, 3

E_INVALID, E_UNSUPPORTED, E_WORKSPACE, E_HIP = -1, -2, -3, -4

_fp = C.c_void_p  # device pointers travel as integers


class Dist(C.Structure):
    _fields_ = [("kind", C.c_int32), ("k", C.c_int32), ("loc", _fp), ("scale", _fp), ("w", _fp),
                ("p0", C.c_float), ("p1", C.c_float), ("p2", C.c_float), ("p3", C.c_float), ("clip", C.c_float), ("aux", _fp)]


class TimeEmbed(C.Structure):
    _fields_ = [("coeff", _fp), ("phase", _fp), ("w", _fp * 4), ("b", _fp * 4), ("n_hidden", C.c_int32),
                ("dim_out", C.c_int32), ("w_out", _fp), ("b_out", _fp)]


class Net(C.Structure):
    _fields_ = [("ctrl_kind", C.c_int32), ("reserved", C.c_int32),
                ("w_in", _fp), ("b_in", _fp), ("w_h1", _fp), ("b_h1", _fp), ("w_h2", _fp), ("b_h2", _fp),
                ("w_out", _fp), ("b_out", _fp), ("t_embed", TimeEmbed), ("score_model", TimeEmbed),
                ("clip_model", C.c_float), ("clip_score", C.c_float), ("scale_score", C.c_float), ("reserved_f", C.c_float)]


class Ref(C.Structure):
    _fields_ = [("kind", C.c_int32), ("k", C.c_int32), ("means_init", _fp), ("vars_init", _fp), ("weights", _fp), ("eigvecs", _fp),
                ("shared_var", C.c_int32), ("reserved", C.c_int32)]


class Adjoint(C.Structure):  # sdeng_adjoint (include/sdeng.h)
    _fields_ = [(n, C.c_void_p) for n in ("xs", "noise", "w", "lam_in", "lam_out", "a0", "a1", "a2", "d0", "d1", "d2", "dout", "dst")] + [
        ("detach_score", C.c_int32), ("score", C.c_void_p)]


class Desc(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("form", C.c_int32), ("flags", C.c_uint32),
                ("B", C.c_int32), ("d", C.c_int32), ("N", C.c_int32),
                ("particle0", C.c_int64), ("seed", C.c_uint64),
                ("coef", _fp), ("x_in", _fp), ("x_out", _fp), ("rnd_out", _fp), ("xs_out", _fp), ("noise_in", _fp),
                ("net", Net), ("ref", Ref), ("target", Dist), ("ref_dist", Dist), ("prior", Dist),
                ("cmcd_g", C.c_float), ("cmcd_clip", C.c_float), ("workspace", _fp), ("workspace_bytes", C.c_size_t),
                ("ev_start", _fp), ("ev_stop", _fp), ("x0_dist", Dist), ("x0_out", _fp)]


class EngineError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"sdeng error {code}: {msg}")
        self.code = code


_LIB = None
LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsdeng.so")

EXPORTS = ["sdeng_abi_version", "sdeng_last_error", "sdeng_workspace_bytes", "sdeng_simulate", "sdeng_logz",
           "sdeng_logz_workspace_bytes", "sdeng_ctrl_forward", "sdeng_dist_eval", "sdeng_dist_workspace_bytes",
           "sdeng_philox_normal", "sdeng_philox_normal_steps", "sdeng_sample_x0", "sdeng_ctrl_vjp", "sdeng_ctrl_vjp_workspace_bytes",
           "sdeng_langevin_moves", "sdeng_langevin_moves_workspace_bytes", "sdeng_kl_adjoint", "sdeng_kl_adjoint_workspace_bytes"]


def lib() -> C.CDLL:
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -m sde_sampler_lrds_amd.build` "
                          "(there is no CPU fallback for the simulate path)")
    import torch  # noqa: F401 -- before the library: libsdeng.so then binds to the HIP runtime torch has loaded (its own copy), instead
    #                          of bringing /opt/rocm's into the process first; with two runtimes the library's launches find no device
    L = C.CDLL(LIB_PATH)
    L.sdeng_abi_version.restype = C.c_int
    L.sdeng_last_error.restype = C.c_char_p
    L.sdeng_workspace_bytes.restype = C.c_size_t
    L.sdeng_workspace_bytes.argtypes = [C.POINTER(Desc)]
    L.sdeng_simulate.restype = C.c_int
    L.sdeng_simulate.argtypes = [C.POINTER(Desc), C.c_void_p]
    L.sdeng_logz.restype = C.c_int
    L.sdeng_logz.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.sdeng_logz_workspace_bytes.restype = C.c_size_t
    L.sdeng_ctrl_forward.restype = C.c_int
    L.sdeng_ctrl_forward.argtypes = [C.POINTER(Desc), C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
    L.sdeng_dist_eval.restype = C.c_int
    L.sdeng_dist_eval.argtypes = [C.POINTER(Dist), C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_size_t, C.c_void_p]
    L.sdeng_dist_workspace_bytes.restype = C.c_size_t
    L.sdeng_dist_workspace_bytes.argtypes = [C.POINTER(Dist), C.c_int32]
    L.sdeng_philox_normal.restype = C.c_int
    L.sdeng_philox_normal.argtypes = [C.c_uint64, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_uint32, C.c_void_p, C.c_void_p]
    L.sdeng_philox_normal_steps.restype = C.c_int
    L.sdeng_philox_normal_steps.argtypes = [C.c_uint64, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_uint32, C.c_void_p, C.c_void_p]
    L.sdeng_sample_x0.restype = C.c_int
    L.sdeng_sample_x0.argtypes = [C.POINTER(Dist), C.c_uint64, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    L.sdeng_ctrl_vjp.restype = C.c_int
    L.sdeng_ctrl_vjp.argtypes = [C.POINTER(Desc), C.c_int32, C.c_int32] + [C.c_void_p] * 12
    L.sdeng_kl_adjoint.restype = C.c_int
    L.sdeng_kl_adjoint.argtypes = [C.POINTER(Desc), C.POINTER(Adjoint), C.c_void_p]
    L.sdeng_kl_adjoint_workspace_bytes.restype = C.c_size_t
    L.sdeng_kl_adjoint_workspace_bytes.argtypes = [C.POINTER(Desc)]
    L.sdeng_ctrl_vjp_workspace_bytes.restype = C.c_size_t
    L.sdeng_ctrl_vjp_workspace_bytes.argtypes = [C.c_int32, C.c_int32]
    L.sdeng_langevin_moves.restype = C.c_int
    L.sdeng_langevin_moves.argtypes = [C.POINTER(Dist), C.POINTER(Dist), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float] + \
        [C.c_void_p] * 7 + [C.c_uint64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.sdeng_langevin_moves_workspace_bytes.restype = C.c_size_t
    L.sdeng_langevin_moves_workspace_bytes.argtypes = [C.POINTER(Dist), C.POINTER(Dist), C.c_int32]
    if L.sdeng_abi_version() != ABI_VERSION:
        raise ImportError(f"libsdeng.so ABI {L.sdeng_abi_version()} != binding ABI {ABI_VERSION}")
    _LIB = L
    return L


def check(rc: int):
    if rc != 0:
        raise EngineError(rc, lib().sdeng_last_error().decode())


class HipEvents:
    """A pair of raw hipEvent_t for timing the step-loop kernel alone (bench.py roofline leg)."""

    def __init__(self):
        self.hip = C.CDLL("libamdhip64.so")
        self.start, self.stop = C.c_void_p(), C.c_void_p()
        for ev in (self.start, self.stop):
            if self.hip.hipEventCreate(C.byref(ev)) != 0:
                raise RuntimeError("hipEventCreate failed")

    def elapsed_ms(self) -> float:
        if self.hip.hipEventSynchronize(self.stop) != 0:
            raise RuntimeError("hipEventSynchronize failed")
        ms = C.c_float()
        if self.hip.hipEventElapsedTime(C.byref(ms), self.start, self.stop) != 0:
            raise RuntimeError("hipEventElapsedTime failed")
        return ms.value
