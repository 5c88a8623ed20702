#!/usr/bin/env python3
"""Digitise the simulation figures the reference's notebooks STORE (outputs of the reference itself, run by its author
on real CasADi + qpOASES) into tests/golden/notebook_figures.npz.

The reference has no tests and stores no numeric controller output, but its notebooks keep the
PNG figures of the closed-loop simulations they ran:  examples/notebooks/cart_on_track_1D_comparison_of_controllers.ipynb
cells 11 / 22 (move to a point: ReactiveQPController / PseudoInverseController), 36 / 54 (track 0.4 sin(0.3 t), which
leaves the rail: the SetConstraint holds the cart at 0), 61 / 78 (follow the same curve as a PATH with a virtual
variable).  Every one of these axes carries lines the notebook drew at KNOWN values (the speed limits +-0.275, the rail
ends 0 and 1: `'k--'`), and matplotlib's default view limits put a 5 % margin around the plotted time range
[0, dt * 1199] - so pixel rows and columns convert to physical values without reading a single tick label:

    value(row) from the two dashed lines' rows;      t(col) from the axes box and the 5 % margins
                                                     (cross-checked against the extent of the curve itself)

What is written: per figure and curve, one sample per pixel column - t, the value at the centre of the curve's pixels,
and the size of a pixel in t and in value (the resolution of the pin: 0.04 - 0.08 s, 0.006 m/s, 0.016 m).  The tests
(tests/test_figure_pins.py, tests/test_gpu_figure_pins.py) run the same skills through the oracle and the HIP
controllers with the notebooks' own loops and require the simulated curves to pass through these samples to within
a pixel and a half.

Runs only where /root/reference exists (this container); the fixture is data (digitised samples of figures the
reference stores), nothing of the reference's text.     python tests/golden/make_figure_pins.py
"""
import base64
import io
import json
import os

import sys

import numpy as np
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.dirname(os.path.abspath(__file__))
NOTEBOOKS = "/root/reference/examples/notebooks/"
CART = "cart_on_track_1D_comparison_of_controllers.ipynb"
PENDULUM = "double_pendulum_2D_comparison_of_controllers.ipynb"
UR5 = "ur5_transformation_matrix_comparison_of_controllers.ipynb"
UR5_INPUT = "ur5_input_experiment.ipynb"
UR5_DQ = "ur5_dual_quaternion_vs_transformation_matrix.ipynb"
BLACK = (0.0, 0.0, 0.0)
RED, BLUE, GREEN = (255.0, 0.0, 0.0), (0.0, 0.0, 255.0), (0.0, 128.0, 0.0)              # "r", "b", "g"
C0, C1, C2 = (31.0, 119.0, 180.0), (255.0, 127.0, 14.0), (44.0, 160.0, 44.0)         # matplotlib's default cycle
MAX_SPEED, MIN_P, MAX_P = 0.275, 0.0, 1.0                   # cart notebook, cell 6
UR5_HOME_Z = 1.001059                                       # T_fk(UR5_home)[2, 3] of urdf/ur5.urdf, the START of the
#                                                             z curve of the UR5 figure (an input of that simulation;
#                                                             with y0 = 0.19145, x0 = 0 it has the norm 1.0192 the notebook prints)

# figure -> (notebook, cell, dt, ticks, [(axes index, calibration, [(curve name, colour)])])
# calibrations:  ("lines", colour, low value, colour, high value)    two horizontal dashed lines drawn at known values
#                ("extremes", colour, low, high)                     a dashed curve with known extreme values
#                ("line+start", [colours], value, curve, value)      one dashed line and the known first value of a curve
SPEED = ("lines", BLACK, -MAX_SPEED, BLACK, MAX_SPEED)
RAIL = ("lines", BLACK, MIN_P, BLACK, MAX_P)
FIGURES = {
    "qp_point":   (CART, 11, 0.01, 1200, [(1, SPEED, [("dp", RED)])]),
    "pinv_point": (CART, 22, 0.01, 1200, [(1, SPEED, [("dp", RED)])]),
    "qp_traj":    (CART, 36, 0.02, 1200, [(0, RAIL, [("p", C0)]), (1, SPEED, [("dp", RED)])]),
    "pinv_traj":  (CART, 54, 0.02, 1200, [(0, RAIL, [("p", C0)]), (1, SPEED, [("dp", RED)])]),
    "qp_path":    (CART, 61, 0.02, 1200, [(0, RAIL, [("p", C0)]), (1, SPEED, [("dp", RED)])]),
    "pinv_path":  (CART, 78, 0.02, 1200, [(0, RAIL, [("p", C0)]), (1, SPEED, [("dp", RED)])]),
    # double pendulum, ReactiveQPController with the table SetConstraints (general inequality rows): cells 16-19 (to
    # the point (0.75, 0.5); joint speeds with the +-0.5 rad/s lines, tool position with its dashed targets) and
    # cells 36-38 (tracking the circle 0.25 (cos, sin)(0.5 t) + (1, 1): its dashed curves reach 0.75 and 1.25)
    "pend_point_dq": (PENDULUM, 18, 0.01, 800, [(1, ("lines", BLACK, -0.5, BLACK, 0.5), [("dq0", C0), ("dq1", C1)])]),
    "pend_point_p":  (PENDULUM, 19, 0.01, 800, [(0, ("lines", GREEN, 0.5, BLUE, 0.75), [("px", BLUE), ("py", GREEN)])]),
    "pend_track_p":  (PENDULUM, 38, 0.01, 2000, [(0, ("extremes", BLACK, 0.75, 1.25), [("px", BLUE), ("py", GREEN)])]),
    # UR5, PseudoInverseController, norm_2 position error behind six 1-D joint-limit sets, speeds saturated at pi / 5
    # (cells 27-32): x, y, z of the tool, the dashed targets all at 0.5
    # UR5, ReactiveQPController with an input_var (a simulated disturbance from t = 10 s on), multidimensional joint
    # limits, speed limits, solve_initial_problem and the slack warm start (ur5_input_experiment.ipynb cells 7-17);
    # no line at a known value: calibrated from the view limits (data range = start position x0 = 0 ... z0)
    "ur5_qp_input":  (UR5_INPUT, 17, 0.01, 4501, [(0, ("box", 0.0, UR5_HOME_Z), [("x", C0), ("y", C1), ("z", C2)])]),
    # UR5 from home to a frame, error norm on a LOG axis (ur5_dual_quaternion_vs_transformation_matrix.ipynb cells 14-27):
    # the 8-row dual-quaternion deviation Q_dist1 under the ReactiveQPController (blue: visible where it leaves the
    # curves drawn over it, from 1e-9.4 down to the rounding floor); tick labels read off the figure: 10^0 at the top
    # tick, 2 decades per tick.  (The green curve, the PseudoInverseController of that notebook, is NOT a pin: the run
    # starts at UR5_home - the elbow is straight, the 8 x 6 Jacobian has rank 5 - with damping_factor 1e-26, so the
    # first ticks are rounding noise divided by 1e-26 and saturated: which way the arm leaves the singularity depends on
    # the last bit of the linear solver, the reference's curve idles near 1 for five seconds, any other arithmetic
    # idles differently.)
    "ur5_qdist1_e":  (UR5_DQ, 27, 0.008, 1001, [(0, ("logticks", 0, 2), [("qp", C0)])]),
    # (cell 29, the one-row Frobenius-norm deviation T_dist1, is not a pin either: its QP curve is visible only in its
    # tail below 1e-7, where the closed loop of a NORM - no gradient at its zero - amplifies a 1e-13 difference of one
    # tick into 0.3 decades at the end of the run: the oracle happens to retrace the stored tail, the HIP path, equal
    # to the oracle to 1e-13 tick by tick, ends 0.3 decades lower)
    "ur5_pinv_p":    (UR5, 32, 0.01, 1000, [(0, ("line+start", [C0, C1, C2], 0.5, "z", UR5_HOME_Z),
                                             [("x", C0), ("y", C1), ("z", C2)])]),
}


def stored_png(notebook, cell):
    nb = json.load(open(NOTEBOOKS + notebook))
    for out in nb["cells"][cell]["outputs"]:
        if "data" in out and "image/png" in out["data"]:
            raw = base64.b64decode(out["data"]["image/png"])
            im = np.array(Image.open(io.BytesIO(raw)).convert("RGBA")).astype(float)
            a = im[..., 3:] / 255.0
            return im[..., :3] * a + 255.0 * (1.0 - a)          # on white
    # `%matplotlib notebook` cells keep their figure as <img src="data:image/png;base64,..."> inside a text/html output
    import moe_figure_pins
    return moe_figure_pins.html_png(notebook, cell)


def axes_boxes(rgb):
    """(top, bottom, left, right) pixel indices of the axes frames: full-width dark rows between two full-height dark
    columns"""
    dark = rgb.max(axis=2) < 90.0
    colsum = dark.sum(axis=0)
    cols = np.nonzero(colsum > 0.6 * colsum.max())[0]
    left, right = int(cols.min()), int(cols.max())
    width = right - left + 1
    rows = [y for y in range(dark.shape[0]) if dark[y, left:right + 1].sum() > 0.97 * width]
    assert len(rows) % 2 == 0 and len(rows) >= 2, rows
    return [(rows[2 * k], rows[2 * k + 1], left, right) for k in range(len(rows) // 2)]


def colour_weight(rgb, colour):
    """closeness of every pixel to a drawing colour, 1 on the colour ... 0 (for black: darkness, cut below 0.45 so that
    the curves' own colours and anti-aliased rims do not count)"""
    if tuple(colour) == BLACK:
        w = np.clip(1.0 - rgb.max(axis=2) / 255.0, 0.0, 1.0)
        w[w < 0.6] = 0.0               # ("g" is (0, 128, 0): half as dark as black)
        return w
    colour = np.array(colour)
    dist = np.sqrt(((rgb - colour) ** 2).sum(axis=2)) / np.sqrt(((255.0 - colour) ** 2).sum())
    return np.clip(1.0 - dist / 0.6, 0.0, 1.0)


def line_rows(rgb, box, colour):
    """sub-pixel rows (pixel-centre coordinates, top first) of the horizontal dashed lines drawn in `colour`: the rows
    where that colour covers a quarter of the axes' width or more"""
    top, bot, left, right = box
    w = colour_weight(rgb, colour)[top + 2:bot - 1, left + 2:right - 1].copy()
    l0, l1, c0, c1 = legend_box(rgb, box)
    w[max(0, l0 - top - 2):max(0, l1 - top - 1), max(0, c0 - left - 2):max(0, c1 - left - 1)] = 0.0
    prof = w.sum(axis=1)
    strong = prof > 0.25 * (right - left)
    groups, y = [], 0
    while y < len(prof):
        if strong[y]:
            y1 = y
            while y1 + 1 < len(prof) and strong[y1 + 1]:
                y1 += 1
            lo, hi = max(0, y - 1), min(len(prof) - 1, y1 + 1)
            ww = prof[lo:hi + 1]
            groups.append(float((ww * (np.arange(lo, hi + 1) + 0.5)).sum() / ww.sum()) + top + 2)
            y = y1 + 1
        else:
            y += 1
    return groups


def extreme_rows(rgb, box, colour):
    """centre rows of the highest and the lowest point of a dashed curve drawn in `colour`: per pixel column the
    centre of the topmost / bottommost run of that colour, then the extreme over the columns"""
    top, bot, left, right = box
    w = colour_weight(rgb, colour).copy()
    l0, l1, c0, c1 = legend_box(rgb, box)
    w[l0:l1 + 1, c0:c1 + 1] = 0.0
    rows = np.arange(rgb.shape[0]) + 0.5
    hi_row, lo_row = np.inf, -np.inf
    for c in range(left + 3, right - 2):
        col = w[top + 2:bot - 1, c]
        ys = np.nonzero(col > 0.0)[0]
        if len(ys) == 0:
            continue
        for first, sign in ((ys[0], 1), (ys[-1], -1)):
            run = [first]
            while 0 <= run[-1] + sign < len(col) and col[run[-1] + sign] > 0.0 and len(run) < 4:
                run.append(run[-1] + sign)
            run = np.array(run)
            centre = float((col[run] * (rows[run + top + 2])).sum() / col[run].sum())
            if sign > 0:
                hi_row = min(hi_row, centre)
            else:
                lo_row = max(lo_row, centre)
    return hi_row, lo_row


def coloured_extent(rgb, box):
    """first and last pixel column of the axes in which ANY coloured curve is drawn (a curve hidden under another one
    is covered by that one): the plotted time range"""
    top, bot, left, right = box
    sat = (rgb.max(axis=2) - rgb.min(axis=2)) > 60.0
    sat[:top + 2] = False
    sat[bot - 1:] = False
    l0, l1, c0, c1 = legend_box(rgb, box)
    sat[l0:l1 + 1, c0:c1 + 1] = False
    cols = np.nonzero(sat[:, left + 2:right - 1].any(axis=0))[0] + left + 2
    return int(cols[0]), int(cols[-1])


def tick_rows(rgb, box):
    """rows (pixel centres) of the tick marks on the left spine"""
    top, bot, left, right = box
    dark = rgb.max(axis=2) < 110.0
    rows = []
    for y in range(top - 2, bot + 3):
        c, run = left - 1, 0
        while c >= 0 and dark[y, c]:
            run += 1
            c -= 1
        if run >= 2:
            rows.append(y + 0.5)
    return rows


def legend_box(rgb, box):
    """the legend's frame inside an axes: its light-grey top and bottom borders are the only long horizontal grey runs
    (the anti-aliased rims of the dashes are a few pixels long)"""
    top, bot, left, right = box
    r, g, b = rgb[..., 0], rgb[..., 1], rgb[..., 2]
    grey = (np.abs(r - g) < 6) & (np.abs(g - b) < 6) & (r > 190) & (r < 236)
    found = []
    for y in range(top + 2, bot - 1):
        xs = np.nonzero(grey[y, left + 2:right - 1])[0]
        if len(xs) < 30:
            continue
        runs = np.split(xs, np.nonzero(np.diff(xs) > 1)[0] + 1)
        longest = max(runs, key=len)
        if len(longest) >= 30:
            found.append((y, int(longest[0]) + left + 2, int(longest[-1]) + left + 2))
    assert found, "no legend frame"
    return (min(f[0] for f in found) - 1, max(f[0] for f in found) + 1,
            min(f[1] for f in found) - 4, max(f[2] for f in found) + 4)


def trace(rgb, box, colour, avoid_rows=()):
    """per pixel column the centre row of the curve drawn in `colour` (followed from the left by continuity: the
    legend holds a short sample of the same colour elsewhere in the axes)"""
    top, bot, left, right = box
    colour = np.array(colour)
    dist = np.sqrt(((rgb - colour) ** 2).sum(axis=2)) / np.sqrt(((255.0 - colour) ** 2).sum())
    # distance of every pixel from the segment white ... colour (the anti-aliased rim of the curve lies ON it; the
    # greys of an anti-aliased black dash do not)
    d, u = 255.0 - rgb, 255.0 - colour
    a = np.clip((d * u).sum(axis=2) / (u * u).sum(), 0.0, 1.0)
    off_mix = np.sqrt(((d - a[..., None] * u) ** 2).sum(axis=2))
    mask = (dist < 0.45) & (off_mix < 60.0)
    mask[:top + 2] = False
    mask[bot - 1:] = False
    l0, l1, c0, c1 = legend_box(rgb, box)
    mask[l0:l1 + 1, c0:c1 + 1] = False        # (the legend's sample of the colour; a curve passing behind the frame
    #                                            loses those columns)
    drawn = np.nonzero(mask[:, left + 2:right - 1].any(axis=0))[0] + left + 2
    extent = (int(drawn[0]), int(drawn[-1]))   # columns where the colour is drawn at all: the plotted time range
    for r in avoid_rows:                       # horizontal dashed lines (possibly of the curve's own colour): the
        mask[int(np.floor(r - 2.0)):int(np.ceil(r + 2.0)) + 1] = False      # curve is not followed across them
    cols, centre, half = [], [], []
    prev, first_col = None, None
    for c in range(left + 2, right - 1):
        ys = np.nonzero(mask[:, c])[0]
        if len(ys) == 0:
            continue
        # contiguous runs (a gap of up to 3 rows is bridged: the dashed reference is drawn over the position line)
        runs, start = [], ys[0]
        for a, b in zip(ys[:-1], ys[1:]):
            if b - a > 4:
                runs.append((start, a))
                start = b
        runs.append((start, ys[-1]))
        if prev is None:
            # the curve starts at the left end of the plotted range (a vertical stroke there - the notebooks leave
            # sample 0 of some arrays at zero - is not a value: wait for the first column with a single short run)
            run = max(runs, key=lambda r: r[1] - r[0])
            if run[1] - run[0] > 7 or len(runs) > 1:
                if first_col is None:
                    first_col = c
                continue
        else:
            run = min(runs, key=lambda r: abs(0.5 * (r[0] + r[1] + 1) - prev))
            if abs(0.5 * (run[0] + run[1] + 1) - prev) > 12.0 + (run[1] - run[0]):
                continue                                     # only the legend's sample in this column
        prev = 0.5 * (run[0] + run[1] + 1)
        cols.append(c)
        centre.append(prev)
        half.append(0.5 * (run[1] - run[0] + 1))
    cols, centre, half = np.array(cols), np.array(centre), np.array(half)
    # second pass: the thresholded runs jitter by a row where other artists cross the line (the dashed reference and
    # the dashed rail ends are drawn over the position line).  Around the running median of the first pass, take the
    # centroid of the "closeness to the colour" weights instead.
    weight = np.clip(1.0 - dist / 0.6, 0.0, 1.0)
    weight[~np.isfinite(weight)] = 0.0
    weight[:top + 2] = 0.0
    weight[bot - 1:] = 0.0
    weight[l0:l1 + 1, c0:c1 + 1] = 0.0
    rows = np.arange(rgb.shape[0]) + 0.5
    keep, refined = [], []
    for k, c in enumerate(cols):
        lo, hi = max(0, k - 4), min(len(cols), k + 5)
        guess = np.median(centre[lo:hi])
        if abs(centre[k] - guess) > 2.5 + 2.0 * np.abs(np.diff(centre[lo:hi])).max(initial=0.0):
            continue                                       # the first pass left the curve in this column
        reach = half[k] + 1.5
        if half[k] > 3.0:
            continue                                       # a jump drawn as a vertical stroke: no value in this column
        band = np.abs(rows - centre[k]) <= reach
        w = weight[:, c] * band
        if w.sum() < 0.8:
            continue
        # columns where another artist (a dash of the reference or of a rail end) lies inside the band are left out
        foreign = band & (off_mix[:, c] > 60.0)
        if foreign.any():
            continue
        keep.append(k)
        refined.append(float((w * rows).sum() / w.sum()))
    keep = np.array(keep)
    return cols[keep], np.array(refined), half[keep], extent


def main():
    out = {}
    for name, (notebook, cell, dt, n_ticks, axes) in FIGURES.items():
        rgb = stored_png(notebook, cell)
        boxes = axes_boxes(rgb)
        t_max = dt * (n_ticks - 1)
        for ax, calib, curves in axes:
            box = boxes[ax]
            top, bot, left, right = box
            # x: matplotlib's default limits are the data range widened by 5 % on either side; the spines (one pixel
            # wide) are centred on the limits
            x_lo, x_hi = left + 0.5, right + 0.5
            per_col = 1.1 * t_max / (x_hi - x_lo)
            # y: two rows with known values
            avoid = []
            if calib[0] == "lines":
                _, colour_lo, v_lo, colour_hi, v_hi = calib
                if colour_lo == colour_hi:
                    rows_found = line_rows(rgb, box, colour_lo)
                    assert len(rows_found) == 2, (name, "expected two dashed lines", rows_found)
                    row_hi, row_lo = rows_found
                else:
                    (row_hi,), (row_lo,) = line_rows(rgb, box, colour_hi), line_rows(rgb, box, colour_lo)
                avoid = [row_hi, row_lo] if colour_lo != BLACK else []
            elif calib[0] == "extremes":
                _, colour, v_lo, v_hi = calib
                row_hi, row_lo = extreme_rows(rgb, box, colour)
            elif calib[0] == "logticks":
                # logarithmic axis (set_yscale("log")): the major tick marks left of the spine are a whole number of
                # decades apart; which decades - the exponent at the top tick and the step - is read off the stored
                # figure's labels BY EYE and written in the table above; a line through the tick rows gives decades
                # per row.  The digitised values are log10 of the plotted error.
                _, top_exp, step = calib
                ticks = tick_rows(rgb, box)
                exps = top_exp - step * np.arange(len(ticks))
                slope, icpt = np.polyfit(np.array(ticks), exps, 1)
                assert np.abs(slope * np.array(ticks) + icpt - exps).max() < 0.6 * abs(slope), "tick rows not evenly spaced"
                row_hi, row_lo = ticks[0], ticks[-1]
                v_hi, v_lo = float(slope * row_hi + icpt), float(slope * row_lo + icpt)
            elif calib[0] == "box":
                # no line at a known value in these axes: the view limits themselves - matplotlib's default is the
                # data range widened by 5 % (as for the time axis) - and the data range is known from the INPUTS of
                # the run (the tool's start position); the spines are centred on the limits
                _, v_min, v_max = calib
                span = v_max - v_min
                v_lo, v_hi = v_min - 0.05 * span, v_max + 0.05 * span
                row_hi, row_lo = top + 0.5, bot + 0.5
            else:
                _, colours, v_lo, start_curve, v_hi = calib
                found = sorted(set(round(r, 1) for col in colours for r in line_rows(rgb, box, col)))
                assert found and max(found) - min(found) < 1.0, (name, "the dashed targets should share a row", found)
                row_lo = float(np.mean(found))
                avoid = [row_lo]
            traced = {curve: trace(rgb, box, colour, avoid) for curve, colour in curves}
            if calib[0] == "line+start":
                row_hi = float(np.mean(traced[start_curve][1][:2]))     # the curve's first two columns
            per_row = (v_hi - v_lo) / (row_lo - row_hi)                  # value per pixel row
            if calib[0] == "line+start":
                # cross-check of the "box" calibration used for ur5_qp_input on a figure that HAS a known line: the view
                # limits put the 0.5 line where it is found
                span = UR5_HOME_Z - 0.0
                predicted = (top + 0.5) + (UR5_HOME_Z + 0.05 * span - v_lo) / (1.1 * span) * (bot - top)
                print("%-13s view-limit calibration puts the dashed targets at row %.2f, found at %.2f" % (name, predicted, row_lo))
                assert abs(predicted - row_lo) < 0.75
            # cross-check of the x calibration against a reference-drawn extent: the curves span [0, t_max]
            first_col, last_col = coloured_extent(rgb, box)
            t_first, t_last = (-0.05 * t_max + (np.array([first_col, last_col]) + 0.5 - x_lo) * per_col)
            assert abs(t_first) < 3.0 * per_col and abs(t_last - t_max) < 3.0 * per_col, (name, t_first, t_last)
            for curve, _ in curves:
                cols, centre, half, extent = traced[curve]
                t = -0.05 * t_max + (cols + 0.5 - x_lo) * per_col
                v = v_hi - (centre - row_hi) * per_row
                keep = (t >= 0.0) & (t <= t_max)
                key = "%s_%s" % (name, curve)
                out[key + "_t"], out[key + "_v"] = t[keep], v[keep]
                out[key + "_half_rows"] = half[keep]
                out[key + "_pixel"] = np.array([per_col, per_row])
                print("%-13s %-3s %3d samples  t in [%.3f, %.3f] of [0, %.2f]  pixel = %.4f s x %.5f  rows %.2f / %.2f"
                      "  range [%.4f, %.4f]" % (name, curve, keep.sum(), t_first, t_last, t_max, per_col, per_row, row_hi,
                                               row_lo, v[keep].min(), v[keep].max()))
        out[name + "_dt"] = np.array(dt)
    # ur5_moe2016_example2.ipynb cells 13-27 (html-embedded figures; interval pins, layout calibration)
    import moe_figure_pins
    moe_figure_pins.collect(out)
    # ur5_dual_quaternion_comparison_of_controllers.ipynb cells 19, 20, 41, 42 (html-embedded, log axes)
    moe_figure_pins.collect_dq(out)
    # ... cells 22-27 and 44-50: the frame_3d figures of the QP and pinv runs (tool path and frame-axis tips in 3-D)
    import frame3d_pins
    frame3d_pins.collect_frames(out, moe_figure_pins.html_png, stored_png)
    frame3d_pins.collect_joint_figure(out, stored_png)      # ... and cell 31 of the UR5 notebook: joints of the pinv point run
    path = os.path.join(HERE, "notebook_figures.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, len(out), "arrays,", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
