"""Integration-method factories with the reference's call shape
(casclik/integration_methods.py:11-23): given a state symbol ``x``, a function ``dx_function(x)`` that
returns its rate as an expression, and a step ``dt``, build a ``Function`` ``x -> x_next``.

    f = get_rk4_function(q, lambda q_: controller_rate(q_), dt);  q_next = f(q_now)

These are host-side helpers over ``casclik_amd.sym`` (where the reference uses CasADi); a batch of robots is
integrated on the device by ``PseudoInverseController.rollout_batch(..., method="euler" | "rk4")`` with the
controller itself as the right-hand side (``clik_pinv_rollout_batch_m``).
"""
from . import sym as cs


def get_euler_function(x, dx_function, dt):
    """explicit Euler step  x + dx(x) dt  (integration_methods.py:11-14)"""
    return cs.Function("feuler", [x], [x + dx_function(x) * dt])


def get_rk4_function(x, dx_function, dt):
    """classical Runge-Kutta step (integration_methods.py:17-23)"""
    k1 = dx_function(x)
    k2 = dx_function(x + (dt / 2.0) * k1)
    k3 = dx_function(x + (dt / 2.0) * k2)
    k4 = dx_function(x + dt * k3)
    x_end = x + (dt / 6.0) * (k1 + 2 * k2 + 2 * k3 + k4)
    return cs.Function("frk4", [x], [x_end])
