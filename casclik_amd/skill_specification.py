"""SkillSpecification: the ordered constraint list plus variable bookkeeping
that every controller is constructed from.

API parity with reference casclik/skill_specification.py:23-250 - same ctor
keywords, ``n_robot_var / n_virtual_var / n_input_var / n_slack_var``,
``slack_var``, ``_has_virtual / _has_input``, ``print_constraints`` and
``count_constraints``.  The constraint list is **stably sorted by priority**
(reference :139-142) - that order is the task stack the device kernels consume.
"""
from __future__ import annotations

import sys

from . import sym as cs
from .constraints import (EqualityConstraint, SetConstraint,
                          VelocityEqualityConstraint, VelocitySetConstraint)


class SkillSpecification(object):
    """Specification of a skill to be executed on the robot.

    Args:
        label (str): name of the skill
        time_var (MX.sym): symbol for time
        robot_var (MX.sym): controllable robot variables
        robot_vel_var (MX.sym): their velocities (created when omitted)
        virtual_var (MX.sym): internal virtual variables
        virtual_vel_var (MX.sym): their velocities (created when omitted)
        input_var (MX.sym): input variables (never differentiated)
        constraints (list): constraint objects
    """

    def __init__(self, label, time_var, robot_var, robot_vel_var=None,
                 virtual_var=None, virtual_vel_var=None, input_var=None,
                 constraints=None):
        self._constraints = []
        self._virtual_var = None
        self._input_var = None
        self._has_virtual = False
        self._has_input = False
        self.label = label
        self.time_var = time_var
        self.robot_var = robot_var
        self.robot_vel_var = robot_vel_var
        self.virtual_var = virtual_var
        self.virtual_vel_var = virtual_vel_var
        self.input_var = input_var
        self.constraints = [] if constraints is None else constraints

    # -- variables -------------------------------------------------------
    @property
    def robot_var(self):
        return self._robot_var

    @robot_var.setter
    def robot_var(self, var):
        self._robot_var = var
        self.n_robot_var = var.size()[0] if var is not None else 0

    @property
    def robot_vel_var(self):
        return self._robot_vel_var

    @robot_vel_var.setter
    def robot_vel_var(self, var):
        if var is None:
            self._robot_vel_var = cs.MX.sym("robot_vel_var", self.n_robot_var)
            return
        if not isinstance(var, cs.MX):
            raise TypeError("robot_vel_var must be cs.MX.sym.")
        if var.size() != self.robot_var.size():
            raise ValueError("robot_var and robot_vel_var must have the same "
                             "dimensions")
        self._robot_vel_var = var

    @property
    def virtual_var(self):
        return self._virtual_var

    @virtual_var.setter
    def virtual_var(self, var):
        self._virtual_var = var
        self.n_virtual_var = var.size()[0] if var is not None else 0
        self._check_var_existence()

    @property
    def virtual_vel_var(self):
        return self._virtual_vel_var

    @virtual_vel_var.setter
    def virtual_vel_var(self, var):
        if var is None:
            self._virtual_vel_var = cs.MX.sym("virtual_vel_var",
                                              self.n_virtual_var)
            return
        if not isinstance(var, cs.MX):
            raise TypeError("virtual_vel_var must be cs.MX.sym.")
        if var.size() != self.virtual_var.size():
            raise ValueError("virtual_vel_var and virtual_var must have the "
                             "same dimensions")
        self._virtual_vel_var = var

    @property
    def input_var(self):
        return self._input_var

    @input_var.setter
    def input_var(self, var):
        self._input_var = var
        self.n_input_var = var.size()[0] if var is not None else 0
        self._check_var_existence()

    # -- constraints -----------------------------------------------------
    @property
    def constraints(self):
        return self._constraints

    @constraints.setter
    def constraints(self, cnstr_list):
        # sorted() is stable: equal priorities keep their insertion order
        self._constraints = sorted(cnstr_list, key=lambda c: c.priority)
        n_slack = 0
        for cnstr in self._constraints:
            if cnstr.constraint_type == "soft":
                n_slack += cnstr.expression.size()[0]
        self.n_slack_var = n_slack
        self.slack_var = cs.MX.sym("slack_var", n_slack) if n_slack else None
        self._check_var_existence()

    def _check_var_existence(self):
        """Set ``_has_virtual`` / ``_has_input``: does any constraint
        expression, target, bound or gain reference the variable?
        (reference :154-200 asks the same through Jacobian sparsity)."""
        def used(var):
            if var is None:
                return False
            for cnstr in self._constraints:
                parts = [cnstr.expression]
                for attr in ("target", "set_min", "set_max", "gain"):
                    val = getattr(cnstr, attr, None)
                    if isinstance(val, cs.MX):
                        parts.append(val)
                if any(cs.depends_on(p, var) for p in parts):
                    return True
            return False
        self._has_virtual = used(self._virtual_var)
        self._has_input = used(self._input_var)

    # -- reporting -------------------------------------------------------
    def print_constraints(self):
        w = sys.stdout.write
        w("SkillSpecification: " + self.label + "\n")
        for cnstr_id, cnstr in enumerate(self.constraints):
            w("#" + str(cnstr_id) + ": " + cnstr.label + "\n")
        w("Has virtual var: " + str(self._has_virtual) + "\n")
        w("Has input var: " + str(self._has_input) + "\n")
        cnt = self.count_constraints()
        w("N constraints: " + str(cnt["all"]) + "\n")
        w("N equality:\n")
        w("\tPos:" + str(cnt["equality"]))
        w("\tVel:" + str(cnt["velocity_equality"]) + "\n")
        w("N set:\n")
        w("\tPos:" + str(cnt["set"]))
        w("\tVel:" + str(cnt["velocity_set"]) + "\n")
        sys.stdout.flush()

    def count_constraints(self):
        cnt = {"all": len(self.constraints), "equality": 0,
               "velocity_equality": 0, "set": 0, "velocity_set": 0,
               "hard": 0, "soft": 0}
        kinds = ((EqualityConstraint, "equality"), (SetConstraint, "set"),
                 (VelocityEqualityConstraint, "velocity_equality"),
                 (VelocitySetConstraint, "velocity_set"))
        for cnstr in self.constraints:
            if cnstr.constraint_type in ("hard", "soft"):
                cnt[cnstr.constraint_type] += 1
            for klass, key in kinds:
                if isinstance(cnstr, klass):
                    cnt[key] += 1
                    break
        return cnt
