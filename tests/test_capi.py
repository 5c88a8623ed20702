"""The C-ABI library: loads, exports every symbol of include/clik.h, struct
layouts agree between C and ctypes, host-only entry points work (CPU only)."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from casclik_amd import _capi, skills
from casclik_amd.lowering import lower_skill

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "clik.h")


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_capi.LIB_PATH):
        from casclik_amd.build import build_hip
        build_hip()
    return _capi.load_library()


def test_exports_every_declared_symbol(lib):
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = sorted(set(re.findall(r"\b(clik_[a-z_0-9]+)\s*\(", text)))
    assert len(declared) >= 16
    for name in declared:
        assert hasattr(lib, name), "libclik_hip.so does not export %s" % name
    assert sorted(_capi.exported_symbols()) == declared
    assert lib.clik_abi_version() == _capi.ABI_VERSION


def test_struct_layouts_match_the_header(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "clik.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                   'sizeof(clik_joint),sizeof(clik_row),sizeof(clik_task),sizeof(clik_skill_desc),'
                   'sizeof(clik_pinv_opts),sizeof(clik_qp_opts),offsetof(clik_skill_desc,rows),offsetof(clik_task,gain));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    want = [C.sizeof(_capi.clik_joint), C.sizeof(_capi.clik_row), C.sizeof(_capi.clik_task),
            C.sizeof(_capi.clik_skill_desc), C.sizeof(_capi.clik_pinv_opts), C.sizeof(_capi.clik_qp_opts),
            _capi.clik_skill_desc.rows.offset, _capi.clik_task.gain.offset]
    assert got == want


def test_shape_describe_and_generated_table(lib):
    """The AOT shape table must contain exactly what the run-time dispatcher
    derives for the BASELINE skills (tools/gen_shapes.py keeps them in sync)."""
    gen = open(os.path.join(ROOT, "casclik_amd", "csrc", "clik_shapes_gen.hpp")).read()
    for name, spec, extra in [("kStackIiwa", skills.stack_skill(), skills.STACK_OPTIONS),
                              ("kPose6Iiwa", skills.pose_skill(), None),
                              ("kPos3Iiwa", skills.position_skill(), None)]:
        opts = {"feedforward": True, "multidim_sets": False, "converge_final_set_to_max": False,
                "pinv_method": "damped", "damping_factor": 1e-7}
        opts.update(extra or {})
        desc = _capi.desc_to_c(lower_skill(spec))
        buf = C.create_string_buffer(4096)
        rc = lib.clik_shape_describe(C.byref(desc), C.byref(_capi.pinv_opts_to_c(opts)), buf, len(buf))
        assert rc == 1
        assert "ShapeDesc %s = %s;" % (name, buf.value.decode()) in gen


def test_create_rejects_malformed_descriptors(lib):
    desc = _capi.desc_to_c(lower_skill(skills.pose_skill()))
    opts = _capi.pinv_opts_to_c({"feedforward": True, "multidim_sets": False,
                                 "converge_final_set_to_max": False, "pinv_method": "damped",
                                 "damping_factor": 1e-7})
    h = C.c_void_p()
    desc.abi_version = 99
    assert lib.clik_pinv_create(C.byref(desc), C.byref(opts), C.byref(h)) == -1
    assert b"ABI version" in lib.clik_last_error()
    desc.abi_version = _capi.ABI_VERSION
    desc.n_q = 15          # (beyond CLIK_MAX_DOF = 14; 9 ... 14 states get a handle that waits for an instantiated kernel)
    assert lib.clik_pinv_create(C.byref(desc), C.byref(opts), C.byref(h)) == -2
    desc.n_q = 7
    desc.tasks[0].out_row0[0] = 500
    assert lib.clik_pinv_create(C.byref(desc), C.byref(opts), C.byref(h)) == -1
    # multidimensional set without multidim_sets: the reference's NotImplementedError
    sdesc = _capi.desc_to_c(lower_skill(skills.stack_skill()))
    assert lib.clik_pinv_create(C.byref(sdesc), C.byref(opts), C.byref(h)) == -2
    assert b"multidim_sets" in lib.clik_last_error()
    assert lib.clik_pinv_create(None, C.byref(opts), C.byref(h)) == -1
    assert lib.clik_pinv_solve_batch(None, 1, None, None, None, None, None, None, None, None) == -1


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(_capi.ClikLibraryError, match="no CPU fallback|not found"):
        _capi.load_library(str(tmp_path / "libclik_hip.so"))


def test_host_only_handles_answer_queries_and_refuse_to_solve(lib, monkeypatch):
    """clik_pinv_create_host / clik_qp_create_host (include/clik.h): a handle without any device allocation - what
    casclik_amd/jit.py uses to get a skill's image words on a machine without a GPU; every solve entry point refuses it.
    The flag is per call: an ordinary create next to it is NOT host-only (here, without a GPU, it fails in hipMalloc
    instead of succeeding); CLIK_HOST_ONLY=1 still turns a whole process host-only."""
    desc = _capi.desc_to_c(lower_skill(skills.stack_skill()))
    opts = _capi.pinv_opts_to_c({"feedforward": True, "multidim_sets": True, "converge_final_set_to_max": False,
                                 "pinv_method": "damped", "damping_factor": 1e-7})
    h = C.c_void_p()
    assert lib.clik_pinv_create_host(C.byref(desc), C.byref(opts), C.byref(h)) == 0
    buf = (C.c_uint64 * 16384)()
    n = lib.clik_pinv_image_words(h, buf, len(buf))
    assert n > 100 and lib.clik_pinv_n_modes(h) == 2
    assert lib.clik_pinv_solve_batch(h, 4, None, None, None, None, None, None, None, None) == -1
    assert b"host-only" in lib.clik_last_error()
    assert lib.clik_pinv_destroy(h) == 0
    import torch
    if not torch.cuda.is_available():
        h2 = C.c_void_p()
        assert lib.clik_pinv_create(C.byref(desc), C.byref(opts), C.byref(h2)) != 0      # (no device: hipMalloc fails)
    monkeypatch.setenv("CLIK_HOST_ONLY", "1")
    h3 = C.c_void_p()
    assert lib.clik_pinv_create(C.byref(desc), C.byref(opts), C.byref(h3)) == 0 and lib.clik_pinv_destroy(h3) == 0
    monkeypatch.delenv("CLIK_HOST_ONLY")
    from casclik_amd import jit
    qd = _capi.desc_to_c(lower_skill(skills.qp_skill()))
    words = jit.host_image_words(lib, "qp", qd, _capi.qp_opts_to_c(0.001, [1.0] * 7, [1.0] * 6))
    assert words and len(words) > 100


def test_ticket_layout(tmp_path):
    """clik_ticket (resident ticks): 256 bytes, the words the Python layer indexes"""
    src = tmp_path / "tk.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "clik.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                   'sizeof(clik_ticket),offsetof(clik_ticket,in_seq),offsetof(clik_ticket,stop),offsetof(clik_ticket,waves),'
                   'offsetof(clik_ticket,ticks_done),offsetof(clik_ticket,ring_depth),offsetof(clik_ticket,integrate_dt),'
                   'offsetof(clik_ticket,max_speed));return 0;}\n')
    exe = tmp_path / "tk"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    # (int32 words 16: ring_depth; float64 words 9, 10: integrate_dt, max_speed - casclik_amd/controllers/pseudo_inverse.py)
    assert [int(v) for v in subprocess.check_output([str(exe)]).split()] == [256, 0, 4 * 32, 4 * 48, 4 * 49, 4 * 16,
                                                                               8 * 9, 8 * 10]
