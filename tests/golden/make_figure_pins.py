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

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
NOTEBOOK = "/root/reference/examples/notebooks/cart_on_track_1D_comparison_of_controllers.ipynb"
RED, BLUE = (255.0, 0.0, 0.0), (31.0, 119.0, 180.0)        # `c="r"`; the first colour of matplotlib's default cycle
MAX_SPEED, MIN_P, MAX_P = 0.275, 0.0, 1.0                   # cell 6 of the notebook

# figure -> (cell, dt, [(axes index, low line value, high line value, curve name, colour)])
FIGURES = {
    "qp_point":   (11, 0.01, [(1, -MAX_SPEED, MAX_SPEED, "dp", RED)]),
    "pinv_point": (22, 0.01, [(1, -MAX_SPEED, MAX_SPEED, "dp", RED)]),
    "qp_traj":    (36, 0.02, [(0, MIN_P, MAX_P, "p", BLUE), (1, -MAX_SPEED, MAX_SPEED, "dp", RED)]),
    "pinv_traj":  (54, 0.02, [(0, MIN_P, MAX_P, "p", BLUE), (1, -MAX_SPEED, MAX_SPEED, "dp", RED)]),
    "qp_path":    (61, 0.02, [(0, MIN_P, MAX_P, "p", BLUE), (1, -MAX_SPEED, MAX_SPEED, "dp", RED)]),
    "pinv_path":  (78, 0.02, [(0, MIN_P, MAX_P, "p", BLUE), (1, -MAX_SPEED, MAX_SPEED, "dp", RED)]),
}
N_TICKS = 1200


def stored_png(cell):
    nb = json.load(open(NOTEBOOK))
    for out in nb["cells"][cell]["outputs"]:
        if "data" in out and "image/png" in out["data"]:
            raw = base64.b64decode(out["data"]["image/png"])
            im = np.array(Image.open(io.BytesIO(raw)).convert("RGBA")).astype(float)
            a = im[..., 3:] / 255.0
            return im[..., :3] * a + 255.0 * (1.0 - a)          # on white
    raise RuntimeError("cell %d stores no figure" % cell)


def axes_boxes(rgb):
    """(top, bottom, left, right) pixel indices of the axes frames: full-width dark rows between two full-height dark
    columns"""
    dark = rgb.max(axis=2) < 90.0
    colsum = dark.sum(axis=0)
    cols = np.nonzero(colsum > 0.6 * colsum.max())[0]
    left, right = int(cols.min()), int(cols.max())
    width = right - left + 1
    rows = [y for y in range(dark.shape[0]) if dark[y, left:right + 1].sum() > 0.97 * width]
    assert len(rows) % 2 == 0 and len(rows) >= 4, rows
    return [(rows[2 * k], rows[2 * k + 1], left, right) for k in range(len(rows) // 2)]


def dashed_rows(rgb, box):
    """sub-pixel rows (pixel-centre coordinates) of the two black dashed lines of an axes"""
    top, bot, left, right = box
    inner = rgb[top + 2:bot - 1, left + 2:right - 1]
    darkness = np.clip(1.0 - inner.max(axis=2) / 255.0, 0.0, 1.0)
    darkness[darkness < 0.45] = 0.0                      # (colours of the curves and their anti-aliased rims are brighter)
    prof = darkness.sum(axis=1)
    strong = prof > 0.25 * (right - left)
    groups, y = [], 0
    while y < len(prof):
        if strong[y]:
            y1 = y
            while y1 + 1 < len(prof) and strong[y1 + 1]:
                y1 += 1
            lo, hi = max(0, y - 1), min(len(prof) - 1, y1 + 1)
            w = prof[lo:hi + 1]
            groups.append(float((w * (np.arange(lo, hi + 1) + 0.5)).sum() / w.sum()) + top + 2)
            y = y1 + 1
        else:
            y += 1
    assert len(groups) == 2, ("expected two dashed lines", groups)
    return groups          # [row of the HIGH value, row of the LOW value] (rows grow downwards)


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


def trace(rgb, box, colour):
    """per pixel column the centre row of the curve drawn in `colour` (followed from the left by continuity: the
    legend holds a short sample of the same colour elsewhere in the axes)"""
    top, bot, left, right = box
    colour = np.array(colour)
    dist = np.sqrt(((rgb - colour) ** 2).sum(axis=2)) / np.sqrt(((255.0 - colour) ** 2).sum())
    mask = dist < 0.45
    mask[:top + 2] = False
    mask[bot - 1:] = False
    l0, l1, c0, c1 = legend_box(rgb, box)
    mask[l0:l1 + 1, c0:c1 + 1] = False        # (the legend's sample of the colour; a curve passing behind the frame
    #                                            loses those columns)
    cols, centre, half = [], [], []
    prev = None
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
            # the curve starts at the left end of the plotted range; the legend is on the right
            run = max(runs, key=lambda r: r[1] - r[0])
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
    # distance of every pixel from the segment white ... colour (the anti-aliased rim of the curve lies ON it)
    d, u = 255.0 - rgb, 255.0 - colour
    a = np.clip((d * u).sum(axis=2) / (u * u).sum(), 0.0, 1.0)
    off_mix = np.sqrt(((d - a[..., None] * u) ** 2).sum(axis=2))
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
    return cols[keep], np.array(refined), half[keep], (cols[0], cols[-1])


def main():
    out = {}
    for name, (cell, dt, curves) in FIGURES.items():
        rgb = stored_png(cell)
        boxes = axes_boxes(rgb)
        t_max = dt * (N_TICKS - 1)
        for ax, v_lo, v_hi, curve, colour in curves:
            box = boxes[ax]
            top, bot, left, right = box
            row_hi, row_lo = dashed_rows(rgb, box)
            per_row = (v_hi - v_lo) / (row_lo - row_hi)                  # value per pixel row
            # x: matplotlib's default limits are the data range widened by 5 % on either side; the spines (one pixel
            # wide) are centred on the limits
            x_lo, x_hi = left + 0.5, right + 0.5
            per_col = 1.1 * t_max / (x_hi - x_lo)
            cols, centre, half, extent = trace(rgb, box, colour)
            t = -0.05 * t_max + (cols + 0.5 - x_lo) * per_col
            v = v_hi - (centre - row_hi) * per_row
            # cross-check of the x calibration against a reference-drawn extent: the curve spans [0, t_max]
            t_first, t_last = (-0.05 * t_max + (np.array(extent) + 0.5 - x_lo) * per_col)
            assert abs(t_first) < 3.0 * per_col and abs(t_last - t_max) < 3.0 * per_col, (name, curve, t_first, t_last, t_max)
            keep = (t >= 0.0) & (t <= t_max)
            key = "%s_%s" % (name, curve)
            out[key + "_t"], out[key + "_v"] = t[keep], v[keep]
            out[key + "_half_rows"] = half[keep]
            out[key + "_pixel"] = np.array([per_col, per_row])
            print("%-11s %-2s  %3d samples  t in [%.3f, %.3f] of [0, %.2f]  pixel = %.4f s x %.5f  dashed rows %.2f / %.2f"
                  "  range [%.4f, %.4f]" % (name, curve, keep.sum(), t_first, t_last, t_max, per_col, per_row, row_hi, row_lo,
                                           v[keep].min(), v[keep].max()))
        out[name + "_dt"] = np.array(dt)
    path = os.path.join(HERE, "notebook_figures.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, len(out), "arrays,", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
