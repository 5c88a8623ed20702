"""ctypes binding of the C ABI declared in include/clik.h.

The HIP library is the only implementation behind this module: if
``libclik_hip.so`` is missing or fails to load, every controller raises
``ClikLibraryError`` - there is no CPU fallback on the product path.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import lowering as L

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "libclik_hip.so"
LIB_PATH = os.path.join(_HERE, LIB_NAME)

ABI_VERSION = 6


class ClikLibraryError(RuntimeError):
    pass


class ClikError(RuntimeError):
    pass


class clik_joint(C.Structure):
    _fields_ = [("type", C.c_int32), ("q_index", C.c_int32),
                ("R", C.c_double * 9), ("p", C.c_double * 3),
                ("axis", C.c_double * 3)]


class clik_row(C.Structure):
    _fields_ = [("a", C.c_double * L.MAX_DOF), ("b", C.c_double * 3),
                ("g", C.c_double * 9), ("h", C.c_double * 3),
                ("c", C.c_double), ("yc", C.c_double * L.MAX_YTERMS),
                ("yi", C.c_int32 * L.MAX_YTERMS), ("n_y", C.c_int32),
                ("t_slot", C.c_int32), ("flags", C.c_int32),
                ("_pad", C.c_int32)]


class clik_task(C.Structure):
    _fields_ = [("cls", C.c_int32), ("m", C.c_int32), ("soft", C.c_int32),
                ("gain_is_matrix", C.c_int32),
                ("attr_ext", C.c_int32),
                ("out_kind", C.c_int32 * L.MAX_M),
                ("out_row0", C.c_int32 * L.MAX_M),
                ("out_nrows", C.c_int32 * L.MAX_M),
                ("gain", C.c_double * (L.MAX_M * L.MAX_M)),
                ("set_min", C.c_double * L.MAX_M),
                ("set_max", C.c_double * L.MAX_M),
                ("target", C.c_double * L.MAX_M),
                ("slack_weight", C.c_double)]


class clik_skill_desc(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("n_q", C.c_int32),
                ("n_x", C.c_int32), ("n_y", C.c_int32),
                ("n_joints", C.c_int32), ("n_tasks", C.c_int32),
                ("n_rows", C.c_int32), ("n_tslots", C.c_int32),
                ("uses_fk", C.c_int32), ("quat_src", C.c_int32),
                ("quat_yi", C.c_int32 * 4), ("quat", C.c_double * 4),
                ("joints", clik_joint * L.MAX_JOINTS),
                ("tasks", clik_task * L.MAX_TASKS),
                ("rows", clik_row * L.MAX_ROWS)]


class clik_pinv_opts(C.Structure):
    _fields_ = [("feedforward", C.c_int32), ("multidim_sets", C.c_int32),
                ("converge_final_set_to_max", C.c_int32),
                ("pinv_method", C.c_int32), ("damping_factor", C.c_double)]


class clik_qp_opts(C.Structure):
    _fields_ = [("weight_shifter", C.c_double),
                ("state_weights", C.c_double * L.MAX_DOF),
                ("slack_weights", C.c_double * L.MAX_QPROWS),
                ("max_iter", C.c_int32), ("_pad", C.c_int32)]


def desc_to_c(d):
    """SkillDescriptor (lowering.py) -> ctypes clik_skill_desc."""
    out = clik_skill_desc()
    out.abi_version = ABI_VERSION
    out.n_q, out.n_x, out.n_y = d.n_q, d.n_x, d.n_y
    out.n_joints = len(d.joints)
    out.n_tasks = len(d.tasks)
    out.n_rows = len(d.rows)
    out.n_tslots = d.n_tslots
    out.uses_fk = 1 if d.uses_fk else 0
    out.quat_src = d.quat_src
    for k in range(4):
        out.quat_yi[k] = int(d.quat_yi[k])
        out.quat[k] = float(d.quat[k])
    for k, j in enumerate(d.joints):
        cj = out.joints[k]
        cj.type = int(j["type"])
        cj.q_index = int(j["q_index"])
        for i in range(9):
            cj.R[i] = float(j["R"][i])
        for i in range(3):
            cj.p[i] = float(j["p"][i])
            cj.axis[i] = float(j["axis"][i])
    for k, t in enumerate(d.tasks):
        ct = out.tasks[k]
        ct.cls, ct.m, ct.soft = int(t["cls"]), int(t["m"]), int(t["soft"])
        ct.gain_is_matrix = int(t["gain_is_matrix"])
        ct.attr_ext = int(t.get("attr_ext", 0))
        for i in range(L.MAX_M):
            ct.out_kind[i] = int(t["out_kind"][i])
            ct.out_row0[i] = int(t["out_row0"][i])
            ct.out_nrows[i] = int(t["out_nrows"][i])
            ct.set_min[i] = float(t["set_min"][i])
            ct.set_max[i] = float(t["set_max"][i])
            ct.target[i] = float(t["target"][i])
        # the C side addresses the matrix gain with row stride m
        for i in range(L.MAX_M * L.MAX_M):
            ct.gain[i] = float(t["gain"][i])
        ct.slack_weight = float(t["slack_weight"])
    for k, r in enumerate(d.rows):
        cr = out.rows[k]
        for i in range(L.MAX_DOF):
            cr.a[i] = float(r["a"][i])
        for i in range(3):
            cr.b[i] = float(r["b"][i])
            cr.h[i] = float(r["h"][i])
        for i in range(9):
            cr.g[i] = float(r["g"][i])
        cr.c = float(r["c"])
        for i in range(L.MAX_YTERMS):
            cr.yc[i] = float(r["yc"][i])
            cr.yi[i] = int(r["yi"][i])
        cr.n_y = int(r["n_y"])
        cr.t_slot = int(r["t_slot"])
        cr.flags = int(r["flags"])
    return out


def pinv_opts_to_c(options):
    o = clik_pinv_opts()
    o.feedforward = 1 if options["feedforward"] else 0
    o.multidim_sets = 1 if options["multidim_sets"] else 0
    o.converge_final_set_to_max = 1 if options["converge_final_set_to_max"] else 0
    method = options["pinv_method"]
    if method == "damped":
        o.pinv_method = 0
    elif method == "standard":
        o.pinv_method = 1
    else:
        raise ValueError("pinv_method must be 'damped' or 'standard'")
    o.damping_factor = float(options["damping_factor"])
    return o


def qp_opts_to_c(mu, state_weights, slack_weights, max_iter=0):
    o = clik_qp_opts()
    o.weight_shifter = float(mu)
    sw = np.asarray(state_weights, dtype=float).reshape(-1)
    if sw.size > L.MAX_DOF:
        raise ValueError("too many state weights")
    for i, v in enumerate(sw):
        o.state_weights[i] = float(v)
    kw = np.asarray(slack_weights, dtype=float).reshape(-1)
    if kw.size > L.MAX_QPROWS:
        raise NotImplementedError("more than %d slack variables" % L.MAX_QPROWS)
    for i, v in enumerate(kw):
        o.slack_weights[i] = float(v)
    o.max_iter = int(max_iter)
    return o


_SYMBOLS = [
    "clik_last_error", "clik_abi_version",
    "clik_pinv_create", "clik_pinv_create_host", "clik_pinv_destroy", "clik_pinv_n_modes", "clik_pinv_kernel_name", "clik_pinv_kernel_variant", "clik_pinv_image_words", "clik_pinv_attach_value_kernel", "clik_pinv_attach_resident_kernel", "clik_pinv_resident_waves", "clik_pinv_resident_run", "clik_pinv_resident_run_state", "clik_ticket_feed", "clik_shape_describe", "clik_pinv_attach_kernel",
    "clik_pinv_solve_batch", "clik_pinv_solve_batch_t", "clik_pinv_rollout_batch", "clik_pinv_rollout_batch_x", "clik_pinv_rollout_batch_m",
    "clik_qp_create", "clik_qp_create_host", "clik_qp_destroy", "clik_qp_n_vars", "clik_qp_n_rows", "clik_qp_workspace_bytes",
    "clik_qp_kernel_name", "clik_qp_shape_describe", "clik_qp_attach_kernel", "clik_qp_image_words", "clik_qp_attach_value_kernel", "clik_qp_is_box_family",
    "clik_qp_attach_resident_kernel", "clik_qp_resident_waves", "clik_qp_resident_run",
    "clik_qp_solve_batch", "clik_qp_solve_batch_hot", "clik_qp_solve_batch_t", "clik_qp_rollout_batch", "clik_qp_rollout_batch_x", "clik_qp_rollout_batch_m", "clik_qp_data_batch",
]

_lib = None


def exported_symbols():
    return list(_SYMBOLS)


def _preload_hip_runtime():
    """libclik_hip.so is linked without a DT_NEEDED on libamdhip64 so that it
    binds to the HIP runtime the process already uses.  Under PyTorch that is
    the copy bundled in torch/lib: load it with RTLD_GLOBAL first."""
    cands = []
    try:
        import torch
        cands.append(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    except Exception:        # pragma: no cover - torch is part of the image
        pass
    cands += ["libamdhip64.so", "/opt/rocm/lib/libamdhip64.so"]
    last = None
    for cand in cands:
        if os.path.isabs(cand) and not os.path.exists(cand):
            continue
        try:
            return C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError as exc:
            last = exc
    raise ClikLibraryError("cannot load the HIP runtime (libamdhip64): %s" % last)


def load_library(path=None):
    """Load libclik_hip.so (once) and declare the prototypes."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    _preload_hip_runtime()
    if not os.path.exists(p):
        raise ClikLibraryError(
            "%s not found - the HIP hot path is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
            "There is no CPU fallback." % p)
    try:
        lib = C.CDLL(p)
    except OSError as exc:
        raise ClikLibraryError("cannot load %s: %s" % (p, exc))
    vp, dp, ip = C.c_void_p, C.c_void_p, C.c_void_p
    lib.clik_last_error.restype = C.c_char_p
    lib.clik_last_error.argtypes = []
    lib.clik_abi_version.restype = C.c_int32
    lib.clik_abi_version.argtypes = []
    lib.clik_pinv_create.restype = C.c_int
    lib.clik_pinv_create.argtypes = [C.POINTER(clik_skill_desc),
                                     C.POINTER(clik_pinv_opts),
                                     C.POINTER(C.c_void_p)]
    lib.clik_pinv_create_host.restype = C.c_int
    lib.clik_pinv_create_host.argtypes = lib.clik_pinv_create.argtypes
    lib.clik_pinv_destroy.restype = C.c_int
    lib.clik_pinv_destroy.argtypes = [vp]
    lib.clik_pinv_n_modes.restype = C.c_int
    lib.clik_pinv_n_modes.argtypes = [vp]
    lib.clik_pinv_kernel_name.restype = C.c_char_p
    lib.clik_pinv_kernel_name.argtypes = [vp]
    lib.clik_pinv_image_words.restype = C.c_int
    lib.clik_pinv_image_words.argtypes = [vp, C.POINTER(C.c_uint64), C.c_int]
    lib.clik_pinv_attach_value_kernel.restype = C.c_int
    lib.clik_pinv_attach_value_kernel.argtypes = [vp, C.c_void_p, C.c_void_p]
    lib.clik_pinv_attach_resident_kernel.restype = C.c_int
    lib.clik_pinv_attach_resident_kernel.argtypes = [vp, C.c_void_p]
    lib.clik_pinv_resident_waves.restype = C.c_int
    lib.clik_pinv_resident_waves.argtypes = [vp, C.c_int64]
    lib.clik_qp_attach_resident_kernel.restype = C.c_int
    lib.clik_qp_attach_resident_kernel.argtypes = [vp, C.c_void_p]
    lib.clik_qp_resident_waves.restype = C.c_int
    lib.clik_qp_resident_waves.argtypes = [vp, C.c_int64]
    lib.clik_qp_resident_run.restype = C.c_int
    lib.clik_qp_resident_run.argtypes = [vp, C.c_int64, C.c_int32, dp, dp, dp, dp, dp, ip, C.c_void_p, C.c_void_p,
                                         C.c_double, C.c_void_p]
    lib.clik_pinv_resident_run_state.restype = C.c_int
    lib.clik_pinv_resident_run_state.argtypes = [vp, C.c_int64, C.c_int32, dp, dp, dp, dp, ip, C.c_void_p, C.c_void_p,
                                                 C.c_double, C.c_double, C.c_double, C.c_void_p]
    lib.clik_pinv_resident_run.restype = C.c_int
    lib.clik_pinv_resident_run.argtypes = [vp, C.c_int64, C.c_int32, dp, dp, dp, dp, ip, C.c_void_p, C.c_void_p,
                                           C.c_double, C.c_void_p]
    lib.clik_ticket_feed.restype = C.c_int
    lib.clik_ticket_feed.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_void_p]
    lib.clik_pinv_kernel_variant.restype = C.c_char_p
    lib.clik_pinv_kernel_variant.argtypes = [vp, C.c_int64]
    lib.clik_pinv_attach_kernel.restype = C.c_int
    lib.clik_pinv_attach_kernel.argtypes = [vp, C.c_void_p, C.c_void_p, C.c_char_p]
    lib.clik_shape_describe.restype = C.c_int
    lib.clik_shape_describe.argtypes = [C.POINTER(clik_skill_desc), C.POINTER(clik_pinv_opts),
                                        C.c_char_p, C.c_int]
    lib.clik_pinv_solve_batch.restype = C.c_int
    lib.clik_pinv_solve_batch.argtypes = [vp, C.c_int64, C.POINTER(C.c_double),
                                          dp, dp, dp, dp, dp, ip, vp]
    lib.clik_pinv_solve_batch_t.restype = C.c_int
    lib.clik_pinv_solve_batch_t.argtypes = [vp, C.c_int64, dp, dp, dp, dp, dp, dp, ip, vp]
    lib.clik_pinv_rollout_batch.restype = C.c_int
    lib.clik_pinv_rollout_batch.argtypes = [vp, C.c_int64, C.c_int32,
                                            C.c_double, C.c_double,
                                            C.POINTER(C.c_double),
                                            dp, dp, dp, ip, vp]
    lib.clik_pinv_rollout_batch_m.restype = C.c_int
    lib.clik_pinv_rollout_batch_m.argtypes = [vp, C.c_int64, C.c_int32, C.c_int32, C.c_double, C.c_double,
                                              C.POINTER(C.c_double), dp, dp, dp, dp, dp, ip, vp]
    lib.clik_pinv_rollout_batch_x.restype = C.c_int
    lib.clik_pinv_rollout_batch_x.argtypes = [vp, C.c_int64, C.c_int32, C.c_double, C.c_double,
                                              C.POINTER(C.c_double), dp, dp, dp, dp, dp, ip, vp]
    lib.clik_qp_create.restype = C.c_int
    lib.clik_qp_create.argtypes = [C.POINTER(clik_skill_desc),
                                   C.POINTER(clik_qp_opts),
                                   C.POINTER(C.c_void_p)]
    lib.clik_qp_create_host.restype = C.c_int
    lib.clik_qp_create_host.argtypes = lib.clik_qp_create.argtypes
    lib.clik_qp_destroy.restype = C.c_int
    lib.clik_qp_destroy.argtypes = [vp]
    lib.clik_qp_kernel_name.restype = C.c_char_p
    lib.clik_qp_kernel_name.argtypes = [vp]
    lib.clik_qp_shape_describe.restype = C.c_int
    lib.clik_qp_shape_describe.argtypes = [C.POINTER(clik_skill_desc), C.c_char_p, C.c_int]
    lib.clik_qp_attach_kernel.restype = C.c_int
    lib.clik_qp_attach_kernel.argtypes = [vp, C.c_void_p, C.c_void_p, C.c_char_p]
    lib.clik_qp_rollout_batch.restype = C.c_int
    lib.clik_qp_rollout_batch.argtypes = [vp, C.c_int64, C.c_int32, C.c_double, C.c_double,
                                          C.POINTER(C.c_double), dp, dp, dp, dp, ip, vp]
    lib.clik_qp_rollout_batch_x.restype = C.c_int
    lib.clik_qp_rollout_batch_x.argtypes = [vp, C.c_int64, C.c_int32, C.c_double, C.c_double,
                                            C.POINTER(C.c_double), dp, dp, dp, dp, dp, dp, ip, vp]
    lib.clik_qp_image_words.restype = C.c_int
    lib.clik_qp_image_words.argtypes = [vp, C.POINTER(C.c_uint64), C.c_int]
    lib.clik_qp_is_box_family.restype = C.c_int
    lib.clik_qp_is_box_family.argtypes = [vp]
    lib.clik_qp_attach_value_kernel.restype = C.c_int
    lib.clik_qp_attach_value_kernel.argtypes = [vp, vp, vp]
    lib.clik_qp_rollout_batch_m.restype = C.c_int
    lib.clik_qp_rollout_batch_m.argtypes = [vp, C.c_int64, C.c_int32, C.c_int32, C.c_double, C.c_double,
                                            C.POINTER(C.c_double), dp, dp, dp, dp, dp, dp, ip, vp]
    lib.clik_qp_n_vars.restype = C.c_int
    lib.clik_qp_n_vars.argtypes = [vp]
    lib.clik_qp_n_rows.restype = C.c_int
    lib.clik_qp_n_rows.argtypes = [vp]
    lib.clik_qp_workspace_bytes.restype = C.c_int64
    lib.clik_qp_workspace_bytes.argtypes = [vp]
    lib.clik_qp_solve_batch.restype = C.c_int
    lib.clik_qp_solve_batch.argtypes = [vp, C.c_int64, C.POINTER(C.c_double),
                                        dp, dp, dp, dp, dp, dp, ip, vp]
    lib.clik_qp_solve_batch_hot.restype = C.c_int
    lib.clik_qp_solve_batch_hot.argtypes = [vp, C.c_int64, C.POINTER(C.c_double),
                                        dp, dp, dp, dp, dp, dp, ip, ip, C.c_int32, vp]
    lib.clik_qp_solve_batch_t.restype = C.c_int
    lib.clik_qp_solve_batch_t.argtypes = [vp, C.c_int64, dp, dp, dp, dp, dp, dp, dp, ip, ip, C.c_int32, vp]
    lib.clik_qp_data_batch.restype = C.c_int
    lib.clik_qp_data_batch.argtypes = [vp, C.c_int64, C.POINTER(C.c_double),
                                       dp, dp, dp, dp, dp, dp, dp, vp]
    if lib.clik_abi_version() != ABI_VERSION:
        raise ClikLibraryError("ABI version mismatch: library %d, python %d"
                               % (lib.clik_abi_version(), ABI_VERSION))
    if path is None:
        _lib = lib
    return lib


CLIK_OK, CLIK_EINVAL, CLIK_EUNSUPPORTED, CLIK_EHIP, CLIK_ENOMEM = 0, -1, -2, -3, -4     # include/clik.h:59-63


def check(lib, rc):
    if rc != 0:
        msg = lib.clik_last_error()
        msg = msg.decode("utf-8", "replace") if msg else "unknown error"
        if rc == -2:
            raise NotImplementedError(msg)
        if rc == -1:
            raise ValueError(msg)
        raise ClikError("clik error %d: %s" % (rc, msg))


def tterms_arg(tterms):
    arr = np.ascontiguousarray(np.asarray(tterms, dtype=np.float64).reshape(-1))
    if arr.size == 0:
        return arr, None
    return arr, arr.ctypes.data_as(C.POINTER(C.c_double))
