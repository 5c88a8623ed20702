"""Constraint objects: the data a skill script hands to a controller.

API parity with the reference classes (reference: casclik/constraints.py -
``EqualityConstraint`` :88-124, ``SetConstraint`` :148-210,
``VelocityEqualityConstraint`` :299-333, ``VelocitySetConstraint`` :336-368):
same constructor signatures, defaults and attribute names (``label``,
``expression``, ``gain``, ``constraint_type``, ``priority``, ``slack_weight``,
``set_min``, ``set_max``, ``target``), same exception types on malformed
input.  Expressions are ``casclik_amd.sym.MX`` instead of ``casadi.MX``; the
derivative helpers answer symbolically through casclik_amd.autodiff, which is
only used for inspection (the controllers lower the expression to the device
task table instead of differentiating graphs per tick).
"""
from __future__ import annotations

import numpy as np

from . import sym as cs

_BIG = 1e10   # default set bound of the reference (constraints.py:199-206)


def _as_mx(expression):
    if isinstance(expression, cs.MX):
        return expression
    return cs.MX(expression)


def _gain_fits(gain, m, label):
    """Gain/expression shape rule of the reference (constraints.py:32-65):
    float always fits; MX/DM must be square with side m or 1; ndarray must be
    m x m; list of numbers must have length m; anything else is a TypeError."""
    if isinstance(gain, (cs.MX, cs.DM)):
        r, c = gain.size()
        return r == c and c in (m, 1)
    if isinstance(gain, np.ndarray):
        return gain.ndim == 2 and gain.shape[0] == gain.shape[1] == m
    if isinstance(gain, float):
        return True
    if isinstance(gain, list):
        for val in gain:
            if not isinstance(val, (float, int)):
                raise TypeError("Unknown gain type in " + label + ". Supported "
                                "are: float, MX, DM, numpy.ndarray, and list "
                                "of floats/ints")
        return len(gain) == m
    # (the text the reference raises, constraints.py:62-64, letter for letter: callers may match on it)
    raise TypeError("Unknown gain type in " + label + "." + "Supported are: float, MX, DM, numpy."
                    + ".ndarray, and list of floats/ints.")


def _bound_fits(bound, m, label, which):
    if isinstance(bound, (float, int)):
        return m == 1
    if isinstance(bound, cs.MX):
        if bound.is_symbolic():
            return False
        return bound.size() == (m, 1)
    if isinstance(bound, cs.DM):
        return bound.size() == (m, 1)
    if isinstance(bound, np.ndarray):
        if bound.ndim == 1:
            return bound.shape[0] == m
        return bound.shape == (m, 1)
    raise TypeError("Unknown " + which + " type in " + label + ". Supported "
                    "are float, MX, DM, and numpy.ndarray")


class BaseConstraint(object):
    """Label + expression + gain (reference: constraints.py:12-85)."""
    constraint_class = "BaseConstraint"

    def __init__(self, label, expression, gain):
        self.label = label
        self.expression = _as_mx(expression)
        self.gain = gain

    def __repr__(self):
        return self.label + "<" + self.constraint_class + " at 0x" + str(id(self)) + ">"

    def size(self):
        return self.expression.size()

    def _check_sizes(self):
        m, cols = self.size()
        if cols != 1:
            return False
        return _gain_fits(self.gain, m, self.label)

    def jacobian(self, var):
        """d expression / d var as an MX (reference: constraints.py:67-73)."""
        from .autodiff import jacobian
        return jacobian(self.expression, var)

    def jtimes(self, varA, varB):
        """(d expression / d varA) * varB (reference: constraints.py:75-80)."""
        return cs.mtimes(self.jacobian(varA), varB)

    def nullspace(self, var):
        """``I - pinv(J) J`` with ``J = d expression / d var`` as an MX (reference: constraints.py:82-85)."""
        J = self.jacobian(var)
        return cs.MX.eye(var.size()[0]) - cs.mtimes(cs.pinv(J), J)


class EqualityConstraint(BaseConstraint):
    """Drive ``expression`` to zero:  J v = -gain*expression - d expr/dt
    (reference: constraints.py:88-124)."""
    constraint_class = "EqualityConstraint"

    def __init__(self, label, expression, gain=1.0, constraint_type="hard",
                 priority=1, slack_weight=1.0):
        BaseConstraint.__init__(self, label, expression, gain)
        self.constraint_type = constraint_type
        self.priority = priority
        self.slack_weight = slack_weight
        if not self._check_sizes():
            raise ValueError("Gain and expression dimensions do not match.")

    def __add__(self, cnstrB):
        return _refuse_sum(self, cnstrB)


def _refuse_sum(left, right):
    """`cnstrA + cnstrB` (constraints.py:126-146, :271-297): the reference checks priority and constraint type and then
    builds the combined gain by MX slice assignment - `gain[:A, :A] = self.gain; gain[:-B, :-B] = cnstrB.gain` with B
    taken from the LEFT operand - whose outcome depends on CasADi's assignment semantics (a scalar fills the block)
    and is not what the docstring promises.  The checks are reproduced; the concatenation itself is not guessed at."""
    if not left.priority == right.priority:
        raise TypeError("Added constraints must have same priority.")
    if not left.constraint_type == right.constraint_type:
        raise TypeError("Added constrains must have same constraint type")
    raise NotImplementedError(
        "adding constraints (%s + %s): the reference assembles the combined gain by MX slice assignment "
        "(constraints.py:139-142) whose result cannot be reproduced without CasADi; build one constraint from "
        "cs.vertcat of the expressions and the gain you mean" % (left.label, right.label))


class SetConstraint(BaseConstraint):
    """Keep ``expression`` inside [set_min, set_max]
    (reference: constraints.py:148-210; unset bounds default to -/+1e10)."""
    constraint_class = "SetConstraint"

    def __init__(self, label, expression, gain=1.0, set_min=None, set_max=None,
                 constraint_type="hard", priority=1, slack_weight=1.0):
        BaseConstraint.__init__(self, label, expression, gain)
        self.constraint_type = constraint_type
        self.priority = priority
        m = self.expression.size()[0]
        self.set_min = -_BIG * np.ones(m) if set_min is None else set_min
        self.set_max = _BIG * np.ones(m) if set_max is None else set_max
        self.slack_weight = slack_weight
        if not self._check_sizes():
            raise ValueError("Gain, set limits, or expression dimensions do "
                             "not match in " + self.label)

    def _check_sizes(self):
        m = self.size()[0]
        ok_gain = BaseConstraint._check_sizes(self)
        ok_min = _bound_fits(self.set_min, m, self.label, "set_min")
        ok_max = _bound_fits(self.set_max, m, self.label, "set_max")
        return ok_gain and ok_min and ok_max

    def __add__(self, cnstrB):
        return _refuse_sum(self, cnstrB)


class VelocityEqualityConstraint(BaseConstraint):
    """Prescribe the time derivative of ``expression``:  J v = target - d expr/dt
    (reference: constraints.py:299-333; sizes are not checked there either)."""
    # (the reference leaves the base class's name on its two velocity constraints: constraints.py:299-368 set no
    # `constraint_class`; kept, since `repr` and callers see it)
    constraint_class = "BaseConstraint"

    def __init__(self, label, expression, gain=1.0, constraint_type="hard",
                 priority=1, target=0.0, slack_weight=1.0):
        BaseConstraint.__init__(self, label, expression, gain)
        self.constraint_type = constraint_type
        self.priority = priority
        self.target = target
        self.slack_weight = slack_weight


class VelocitySetConstraint(BaseConstraint):
    """Bound the time derivative of ``expression``
    (reference: constraints.py:336-368; defaults -/+1e10, no size check)."""
    # (the reference leaves the base class's name on its two velocity constraints: constraints.py:299-368 set no
    # `constraint_class`; kept, since `repr` and callers see it)
    constraint_class = "BaseConstraint"

    def __init__(self, label, expression, gain=1.0, set_min=-_BIG, set_max=_BIG,
                 constraint_type="hard", priority=1, slack_weight=1.0):
        BaseConstraint.__init__(self, label, expression, gain)
        self.constraint_type = constraint_type
        self.priority = priority
        self.set_min = set_min
        self.set_max = set_max
        self.slack_weight = slack_weight
