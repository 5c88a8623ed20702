"""Build-level guards (CPU only): what the compiler made of the shape-specialised
kernels, and what structure the library derives for the BASELINE skills."""
import ctypes as C
import os

import pytest

import casclik_amd as cc
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
        if "occ2" in name:
            # the large-batch build trades a few spilled doubles for two waves per SIMD (measured +11 % at
            # 1 M instances): bounded, and it must really reach occupancy 2
            assert r["ScratchSize"] <= 256 and r["VGPRs"] <= 256 and r["Occupancy"] >= 2, (name, r)
            continue
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
    assert init.count("{1, 2, 3, 4, 5, 6, 7, 0, 0, 0, 0, 0, 0, 0}") == 2
    # ... the pose rows pick single components of p (bits 0..2) and of the orientation error (bits 12..14)
    assert "1u, 2u, 4u, 4096u, 8192u, 16384u" in init
    # 7 sets (128 modes): outside the static family (at most 64 mode bodies per kernel; the mode-scan kernel serves
    # it); with 6 sets the joint-space first equality - a constant Jacobian, processed twice - is inside
    from casclik_amd import sym as cs
    import casclik_amd as cc
    t, q = cs.MX.sym("t"), cs.MX.sym("q", 7)
    for n_sets, inside in ((7, 0), (6, 1)):
        cons = [cc.SetConstraint("s%d" % i, q[i], set_min=-1.0, set_max=1.0, priority=i) for i in range(n_sets)]
        cons.append(cc.EqualityConstraint("c", q - 0.1, priority=9))
        rc4, _ = _describe(lib, cc.SkillSpecification("many", t, q, constraints=cons), dict(opts, multidim_sets=False))
        assert rc4 == inside


def test_instantiated_notebook_qp_kernel_has_no_scratch(tmp_path):
    """The kernel instantiated for the dual-quaternion notebook QP (generated 8-row error, joint limits and
    speed limits merged into six box rows) must stay in registers: 12 separate rows spilled 784 B per lane."""
    import subprocess
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from extern_skills import dual_quaternion_skill
    from casclik_amd import jit
    from casclik_amd.build import parse_resource_remarks, FLAGS
    hipcc = jit._hipcc()
    if hipcc is None:
        pytest.skip("hipcc not available")
    lib = _capi.load_library()
    d = lower_skill(dual_quaternion_skill(skills.ur5(), "Q_dist2"))
    cdesc = _capi.desc_to_c(d)
    buf = C.create_string_buffer(16384)
    assert lib.clik_qp_shape_describe(C.byref(cdesc), buf, len(buf)) == 1
    src = tmp_path / "dq_qp.hip"
    src.write_text(jit._QP_TEMPLATE % {"init": buf.value.decode(), "extern": d.extern_source()})
    out = subprocess.run([hipcc] + FLAGS + ["-c", str(src), "-o", str(tmp_path / "dq_qp.o")],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert out.returncode == 0, out.stdout.decode()[-2000:]
    res = parse_resource_remarks(out.stdout.decode())
    kernels = {k: v for k, v in res.items() if "qp_solve_static" in k or "qp_rollout_static_kernel" in k}
    assert len(kernels) == 4          # per tick (one time / one time per instance), rollout (Euler / Runge-Kutta)
    for name, r in kernels.items():
        assert r["ScratchSize"] == 0, (name, r)


def test_instantiated_kernels_with_expression_attributes_have_no_scratch(tmp_path):
    """gains / bounds given as expressions (ExternAttr code + the per-lane attribute slice of the task cache)
    must dissolve into registers: one run-time evaluated offset pins the whole task cache in scratch"""
    import subprocess
    import numpy as np
    import casclik_amd as cc
    from casclik_amd import jit, sym as cs
    from casclik_amd.build import parse_resource_remarks, FLAGS
    hipcc = jit._hipcc()
    if hipcc is None:
        pytest.skip("hipcc not available")
    lib = _capi.load_library()
    fk = skills.ur5()
    t, q = cs.MX.sym("t"), cs.MX.sym("q", 6)
    gain = 0.5 + 0.3 * cs.sin(0.2 * t) + 0.1 * q[0] * q[0]
    lim = cs.vertcat(*[1.0 + 0.1 * cs.cos(t) for _ in range(6)])
    spec = cc.SkillSpecification("symgain", t, q, constraints=[
        cc.SetConstraint("lims", q, gain=gain, set_min=-lim, set_max=lim, priority=0),
        cc.EqualityConstraint("move", fk["T_fk"](q)[:3, 3] - np.array([0.3, 0.2, 0.4]), gain=gain,
                              constraint_type="soft", priority=1)])
    d = lower_skill(spec)
    cdesc = _capi.desc_to_c(d)
    buf = C.create_string_buffer(16384)
    assert lib.clik_qp_shape_describe(C.byref(cdesc), buf, len(buf)) == 1
    src = tmp_path / "attr_qp.hip"
    src.write_text(jit._QP_TEMPLATE % {"init": buf.value.decode(), "extern": d.extern_source()})
    out = subprocess.run([hipcc] + FLAGS + ["-c", str(src), "-o", str(tmp_path / "attr_qp.o")],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert out.returncode == 0, out.stdout.decode()[-2000:]
    res = parse_resource_remarks(out.stdout.decode())
    kernels = {k: v for k, v in res.items() if "_static_" in k}
    assert len(kernels) == 4
    for name, r in kernels.items():
        assert r["ScratchSize"] == 0, (name, r)


def test_value_specialised_kernels_of_the_config3_skill_have_no_scratch(tmp_path):
    """The kernels casclik_amd/jit.py::attach_values instantiates for BASELINE config 3 with the skill's numbers compiled
    in (four lanes per instance, one lane per instance, their rollouts) must stay in registers, and the one-lane kernel
    must fit two waves per SIMD (that is what makes it the large-batch kernel).  The skill-image words come from a
    host-only handle (CLIK_HOST_ONLY, include/clik.h): no GPU, no recorded fixture."""
    import re
    import subprocess
    from casclik_amd import jit, _capi
    from casclik_amd.build import parse_resource_remarks, FLAGS, CSRC
    from casclik_amd.lowering import lower_skill
    hipcc = jit._hipcc()
    if hipcc is None:
        pytest.skip("hipcc not available")
    spec = skills.stack_skill(skills.iiwa())
    full = cc.PseudoInverseController(skill_spec=spec, options=dict(skills.STACK_OPTIONS)).options
    words = jit.host_image_words(_capi.load_library(), "pinv", _capi.desc_to_c(lower_skill(spec)), _capi.pinv_opts_to_c(full))
    gen = open(os.path.join(CSRC, "clik_shapes_gen.hpp")).read()
    init = re.search(r"kStackIiwa\s*=\s*(\{.*?\});", gen, re.S).group(1)
    text = jit._VALUE_TEMPLATE.replace("%(nwords)d", str(len(words))).replace(
        "%(words)s", ", ".join(w + "ull" for w in words)) % {"init": init, "extern": ""}
    src = tmp_path / "stack_values.hip"
    src.write_text(text)
    # (compiled as shipped: with the scheduling strategy jit.py picks for this translation unit)
    sched = jit.sched_strategy(text, init)
    assert sched == "max-memory-clause"
    out = subprocess.run([hipcc] + FLAGS + ["-DCLIK_VALUE_KERNEL"] + jit.sched_flags(sched) + ["-c", str(src),
                          "-o", str(tmp_path / "stack_values.o")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert out.returncode == 0, out.stdout.decode()[-2000:]
    res = parse_resource_remarks(out.stdout.decode())
    kernels = {k: v for k, v in res.items() if "_static_" in k}
    # team solve, team rollout (Euler, Runge-Kutta: its own instantiation each since round 6), lane solve, lane rollout
    # (Euler, Runge-Kutta)
    assert len(kernels) == 6, sorted(kernels)
    euler = [r for k, r in kernels.items() if "pinv_rollout_static_team_kernel" in k and "ELi1EE" in k]
    assert len(euler) == 1 and euler[0].get("AGPRs", 0) <= 8, euler      # (the Euler rollout carries no Runge-Kutta staging)
    for name, r in kernels.items():
        assert r["ScratchSize"] == 0, (name, r)
        if "pinv_solve_static_values_kernel" in name:
            assert r["VGPRs"] <= 256 and r.get("AGPRs", 0) == 0 and r["Occupancy"] >= 2, (name, r)


def test_value_kernels_of_the_baseline_skills_are_prebuilt():
    """__graft_entry__.build() instantiates the value-specialised kernels of BASELINE configs 2, 3 and 4 ahead of time
    (from the recorded skill-image words) under the cache names the controllers compute at set-up"""
    from casclik_amd import jit
    if jit._hipcc() is None:
        pytest.skip("hipcc not available")
    built = jit.prebuild_value_kernels()
    assert [n for n, _ in built] == ["stack_iiwa", "pose_iiwa", "qp_iiwa"]
    for _, tag in built:
        assert os.path.exists(os.path.join(jit.CACHE, "clik_shape_%s.so" % tag))


def test_value_kernels_of_the_team_family_compile_for_long_input_rows():
    """The value-specialised kernels (launched, rollout AND resident) of a config-3-family skill whose input_var has 14 /
    13 entries (targets for the pose and for every joint): the resident kernel's quad shares its rows two elements per
    lane and needs two rounds for rows longer than eight - a static_assert there once took the whole instantiation (and
    with it the four-lanes kernel of such skills) down, unseen by the BASELINE skills (7 entries).  No GPU needed."""
    import numpy as np
    import casclik_amd as cc
    from casclik_amd import jit, sym as cs
    from casclik_amd.controllers.pseudo_inverse import PseudoInverseController
    if jit._hipcc() is None:
        pytest.skip("no hipcc")
    lib = _capi.load_library()
    for fk in (skills.iiwa(), skills.ur5()):
        n = len(fk["joint_names"])
        t, q, y = cs.MX.sym("t"), cs.MX.sym("q", n), cs.MX.sym("y", 7 + n)
        T = fk["T_fk"](q)
        cons = [cc.EqualityConstraint("joints", cs.vertcat(*[q[j] - y[7 + j] for j in range(n)]), gain=1.0, priority=2),
                cc.SetConstraint(label="limits", expression=q, priority=0, set_min=np.array(fk["lower"]),
                                 set_max=np.array(fk["upper"])),
                cc.EqualityConstraint("task", skills._pose_expression(T, y), gain=3.0, priority=1)]
        spec = cc.SkillSpecification("long_rows", t, q, input_var=y, constraints=cons)
        desc = lower_skill(spec)
        cdesc = _capi.desc_to_c(desc)
        copts = _capi.pinv_opts_to_c(PseudoInverseController(skill_spec=spec, options={"multidim_sets": True}).options)
        ok, init = jit.shape_of(lib, cdesc, copts)
        assert ok and desc.n_y == 7 + n
        words = jit.host_image_words(lib, "pinv", cdesc, copts)
        text = jit._VALUE_TEMPLATE.replace("%(nwords)d", str(len(words))).replace("%(words)s", ", ".join(w + "ull" for w in words))
        so, tag = jit.build_shape_library(init, False, template=text, defines=("-DCLIK_VALUE_KERNEL",), extern="")
        assert so is not None and os.path.exists(so)


def test_scheduling_strategy_per_translation_unit(monkeypatch):
    """casclik_amd/jit.py::sched_strategy (round 6, profiles/r6_sched_ab.txt): QP units max-ilp, value-specialised pinv
    units of skills with SetConstraints max-memory-clause, everything else the compiler's default; CLIK_JIT_SCHED
    overrides; the choice is part of the cache tag and not of the recorded request."""
    import json
    from casclik_amd import jit
    monkeypatch.delenv("CLIK_JIT_SCHED", raising=False)
    stack = "{7, 3, {1, 0, 0, 0, 0, 0, 0, 0}, {7, 6, 7, 0, 0, 0, 0, 0}}"
    pose = "{7, 1, {0, 0, 0, 0, 0, 0, 0, 0}, {6, 0, 0, 0, 0, 0, 0, 0}}"
    assert jit.sched_strategy(jit._QP_VALUE_TEMPLATE, stack) == "max-ilp"
    assert jit.sched_strategy(jit._QP_TEMPLATE, pose) == "max-ilp"
    assert jit.sched_strategy(jit._VALUE_TEMPLATE, stack) == "max-memory-clause"
    assert jit.sched_strategy(jit._VALUE_TEMPLATE, pose) is None
    assert jit.sched_strategy(jit._TEMPLATE, stack) is None
    assert jit.sched_flags(None) == [] and "-amdgpu-use-amdgpu-trackers" in jit.sched_flags("max-ilp")
    assert jit.sched_flags("max-memory-clause") == ["-mllvm", "-amdgpu-sched-strategy=max-memory-clause"]
    monkeypatch.setenv("CLIK_JIT_SCHED", "default")
    assert jit.sched_strategy(jit._QP_VALUE_TEMPLATE, stack) is None
    monkeypatch.setenv("CLIK_JIT_SCHED", "max-ilp")
    assert jit.sched_strategy(jit._TEMPLATE, pose) == "max-ilp"
    monkeypatch.delenv("CLIK_JIT_SCHED")
    # every recorded request carries today's choice into its cache tag, none into its own name
    recs = jit._records()
    assert recs and all("sched" not in json.dumps(meta.get("flags", [])) for _, meta, _ in recs)
    assert {meta["_sched"] for _, meta, _ in recs} == {None, "max-ilp", "max-memory-clause"}
