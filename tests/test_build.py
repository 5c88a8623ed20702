"""Build-level guards (CPU only): what the compiler made of the shape-specialised
kernels, and what structure the library derives for the BASELINE skills."""
import ctypes as C
import os

import pytest

from casclik_amd import _capi, build, skills
from casclik_amd.lowering import lower_skill


@pytest.fixture(scope="module")
def resources():
    if not os.path.exists(build.RESOURCES) or not os.path.exists(_capi.LIB_PATH):
        build.build_hip(force=True)
    return build.kernel_resources()


def test_static_kernels_use_no_scratch(resources):
    """Every per-lane array of the shape-specialised kernels must stay in registers: one
    index that is not a constant expression puts hundreds of bytes per lane in scratch
    (it happened: +15 % tick time, 4x the HBM write traffic).  The compiler's own
    resource report is the guard."""
    static = {k: v for k, v in resources.items() if "static" in k}
    assert len(static) >= 12
    for name, r in static.items():
        assert r["ScratchSize"] == 0, (name, r)
        # (spills may go to AGPRs, which is register space: only memory scratch is forbidden)
        assert r["VGPRs"] + r["AGPRs"] <= 512


def test_single_task_kernels_fit_two_waves_per_simd(resources):
    """The 3-D position kernels are small enough for >= 2 waves per SIMD."""
    for name, r in resources.items():
        if "pinv_solve_static_kernel" in name and "kPos3" in name:
            assert r["Occupancy"] >= 2, (name, r)


def _describe(lib, spec, opts):
    desc = _capi.desc_to_c(lower_skill(spec))
    buf = C.create_string_buffer(16384)
    rc = lib.clik_shape_describe(C.byref(desc), C.byref(_capi.pinv_opts_to_c(opts)), buf, len(buf))
    return rc, buf.value.decode()


def test_shape_of_the_config3_stack():
    lib = _capi.load_library()
    fk = skills.iiwa()
    opts = {"feedforward": True, "multidim_sets": True, "converge_final_set_to_max": False,
            "pinv_method": "damped", "damping_factor": 1e-7}
    rc, init = _describe(lib, skills.stack_skill(fk), opts)
    assert rc == 1
    # joint-limit set and joint centering are joint-space tasks (unit rows q0..q6) ...
    assert init.count("{1, 2, 3, 4, 5, 6, 7, 0, 0, 0, 0, 0}") == 2
    # ... the pose rows pick single components of p (bits 0..2) and of the orientation error (bits 12..14)
    assert "1u, 2u, 4u, 4096u, 8192u, 16384u" in init
    # 6 sets (64 modes): outside the static family (dynamic kernel serves it)
    from casclik_amd import sym as cs
    import casclik_amd as cc
    t, q = cs.MX.sym("t"), cs.MX.sym("q", 7)
    cons = [cc.SetConstraint("s%d" % i, q[i], set_min=-1.0, set_max=1.0, priority=i) for i in range(6)]
    cons.append(cc.EqualityConstraint("c", q - 0.1, priority=9))
    rc4, _ = _describe(lib, cc.SkillSpecification("many", t, q, constraints=cons), dict(opts, multidim_sets=False))
    assert rc4 == 0
