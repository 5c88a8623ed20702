"""casclik_amd - MI355X-native batched CLIK with the casclik API.

Import surface of the reference package (casclik/__init__.py:1-3):
``import casclik_amd as cc`` then ``cc.EqualityConstraint``,
``cc.SkillSpecification``, ``cc.PseudoInverseController`` ...
The CasADi-free expression front-end is ``casclik_amd.sym`` (use it where the
reference scripts say ``import casadi as cs``) and the URDF converter is
``casclik_amd.converter``; ``casclik_amd.casadi_geom`` / ``numpy_geom`` stand in for the
urdf2casadi modules of the same names (quaternion and dual-quaternion helpers).
"""
from casclik_amd.constraints import (BaseConstraint, EqualityConstraint,
                                     SetConstraint, VelocityEqualityConstraint,
                                     VelocitySetConstraint)
from casclik_amd.skill_specification import SkillSpecification
from casclik_amd.controllers import PseudoInverseController, ReactiveQPController
from casclik_amd.urdf import converter
from casclik_amd import sym
from casclik_amd import integration_methods
from casclik_amd.geom import casadi_geom, numpy_geom
# names the reference's `from casclik.controllers import *` / `from casclik.constraints import *` leave on the package
# (casclik/__init__.py:1-3): the controller modules and `cs`
from casclik_amd.controllers import base_controller, pseudo_inverse, reactive_qp
from casclik_amd.controllers.base_controller import ResidentWatchdog
from casclik_amd import sym as cs


class _OutOfScopeController(object):
    """ReactiveNLPController / ModelPredictiveController of the reference (controllers/reactive_nlp.py,
    model_predictive.py: IPOPT-backed) are outside this package's scope - the MI355X hot path is the closed-form
    PseudoInverseController and the ReactiveQPController.  The names exist so that a script that
    merely imports them keeps working; constructing one says so."""
    controller_type = None

    def __init__(self, *args, **kwargs):
        raise NotImplementedError(
            "%s is not part of casclik_amd: the device path covers PseudoInverseController and "
            "ReactiveQPController (the reference's NLP / MPC controllers need an NLP solver)" % self.controller_type)


class ReactiveNLPController(_OutOfScopeController):
    controller_type = "ReactiveNLPController"


class ModelPredictiveController(_OutOfScopeController):
    controller_type = "ModelPredictiveController"

__version__ = "0.1.0"
