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

__version__ = "0.1.0"
