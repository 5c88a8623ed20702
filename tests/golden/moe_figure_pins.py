#!/usr/bin/env python3
"""Digitise the figures examples/notebooks/ur5_moe2016_example2.ipynb STORES (cells 13-27; `%matplotlib notebook` keeps
them as base64 PNGs inside `text/html` outputs) - the reference's only real-CasADi runs of

  * PseudoInverseController(..., options={"multidim_sets": True}) with the multidimensional SetConstraint ACTIVE
    (cell 11, `:429`; pseudo_inverse.py:192-257, 289-298, 352-355), and of
  * three 1-D wall SetConstraints that activate on a 6-DoF arm (8 modes; pseudo_inverse.py:107-190),
  * next to the ReactiveQPController on the same two skills (hard SetConstraint rows with gain 5e2,
    reactive_qp.py:221-225) -

into interval pins (called from make_figure_pins.py; arrays `moe_*` of tests/golden/notebook_figures.npz).

What the figures hold.  Four controllers are drawn into each axes in the order qp, nlp, pinv, mpc (cell 12's printed
order of `controllers.keys()`; the legends agree), tab10 colours, 1.5 pt lines.  A curve drawn EARLIER is visible
only where the later ones leave it.  So a figure pins a controller in two ways:

  union     every coloured pixel of a column, whatever its colour: a hidden curve lies UNDER them with its whole line
            width - its centre can take the values of the band shrunk by half a line width on either side;
  visible   the pixels of the curve's own colour (pinv: green): each lies within half a line width of the centre.

Both are stored as intervals (t of the pixel column's centre, lowest and highest value of the curve's centre) together
with the size of a pixel; tests/golden/figure_skills.py::interval_deviation measures how far a simulated curve misses
them.  The mode figures (cells 17, 18: `mode_sim` of the two PseudoInverseControllers) are stored as levels per pixel
column where the drawn stroke is horizontal, plus the columns a chattering run fills.

Calibration.  Pixel <-> value from the LAYOUT of the stored PNG: matplotlib's default subplot fractions (left 0.125,
right 0.9, bottom 0.11, top 0.88 of the figure; figure sizes 6.4 x 4.8 in and nice_plotting.latexify(3.5, 0.7 * 2.1636)
at 100 dpi = the PNG sizes) and default view limits = the drawn data's range widened by 5 % - every axes used here has
its range set by artists of KNOWN values (walls, the desired trajectory `fpath_des`, the start position, t in [0, 80]).
Straight horizontal / vertical artists are snapped to the pixel grid by Agg, curved ones are not, so each calibration is
cross-checked three ways before anything is written: the dash-dot walls (drawn at known values; snapped: within 0.5 px +
measurement), the spines, and a least-squares fit of the dotted desired trajectory (known for every t, not snapped).

Runs only where /root/reference exists; the fixture is data (digitised samples), nothing of the reference's text.
"""
import base64
import io
import json
import re

import numpy as np
from PIL import Image

NOTEBOOKS = "/root/reference/examples/notebooks/"
MOE = "ur5_moe2016_example2.ipynb"
BLACK = (0.0, 0.0, 0.0)
C0, C1, C2, C3 = (31.0, 119.0, 180.0), (255.0, 127.0, 14.0), (44.0, 160.0, 44.0), (214.0, 39.0, 40.0)   # tab10
PURE_BLUE, PURE_GREEN = (0.0, 0.0, 255.0), (0.0, 128.0, 0.0)                                        # "b", "g"
LINE_W = 1.5 * 100.0 / 72.0          # default line width (1.5 pt) in pixels at 100 dpi
T_END = 80.0                         # cell 12: 10000 ticks of 0.008 s
WALLS = np.array([[0.1, 0.6], [-0.5, 0.4], [-0.3, 0.25]])         # cell 7
HOME_P = np.array([0.22591942, -0.43881418, -0.25264144])        # the tool at UR5_home (DH table of cell 2): the curves' start


def path_des(t):
    """cell 7 (omega = 0.1)"""
    s, c = np.sin(0.1 * t), np.cos(0.1 * t)
    return np.array([0.5 * s * s + 0.2, 0.5 * c + 0.25 * s, 0.5 * s * c + 0.1])


def html_png(notebook, cell):
    """the PNG a `%matplotlib notebook` cell stores: <img src="data:image/png;base64,..."> inside its text/html output"""
    nb = json.load(open(NOTEBOOKS + notebook))
    for out in nb["cells"][cell]["outputs"]:
        if "data" in out and "text/html" in out["data"]:
            m = re.search(r'src="data:image/png;base64,([^"]+)"', "".join(out["data"]["text/html"]))
            if m:
                im = np.array(Image.open(io.BytesIO(base64.b64decode(m.group(1)))).convert("RGBA")).astype(float)
                a = im[..., 3:] / 255.0
                return im[..., :3] * a + 255.0 * (1.0 - a)          # on white
    raise RuntimeError("cell %d stores no html figure" % cell)


class Calib(object):
    """t(x), v(y) for continuous pixel coordinates (pixel column c covers [c, c + 1), row r covers [r, r + 1))"""

    def __init__(self, x_of_t0, px_per_t, y_of_v0, px_per_v):
        self.x0, self.sx, self.y0, self.sy = float(x_of_t0), float(px_per_t), float(y_of_v0), float(px_per_v)

    def t(self, x):
        return (np.asarray(x, float) - self.x0) / self.sx

    def v(self, y):
        return (self.y0 - np.asarray(y, float)) / self.sy

    def x(self, t):
        return self.x0 + np.asarray(t, float) * self.sx

    def y(self, v):
        return self.y0 - np.asarray(v, float) * self.sy

    @property
    def pixel(self):
        return np.array([1.0 / self.sx, 1.0 / self.sy])


def layout_calibration(shape, fig_px, xlim, ylim):
    """default subplot fractions of a figure of fig_px = (width, height) pixels (floats; the canvas is their integer
    part and Agg flips y about the canvas height)"""
    w, h = fig_px
    rows = shape[0]
    x_lo, x_hi = 0.125 * w, 0.9 * w
    y_bot, y_top = rows - 0.11 * h, rows - 0.88 * h
    sx = (x_hi - x_lo) / (xlim[1] - xlim[0])
    sy = (y_bot - y_top) / (ylim[1] - ylim[0])
    return Calib(x_lo - xlim[0] * sx, sx, y_bot + ylim[0] * sy, sy), (y_top, y_bot, x_lo, x_hi)


def margins(lo, hi):
    return lo - 0.05 * (hi - lo), hi + 0.05 * (hi - lo)


def saturated(rgb):
    return (rgb.max(axis=2) - rgb.min(axis=2)) > 60.0


def colour_mask(rgb, colour):
    """pixels on the segment white ... colour, at least half way to the colour (an anti-aliased rim towards ANOTHER colour
    or towards black lies off that segment)"""
    colour = np.array(colour)
    d, u = 255.0 - rgb, 255.0 - colour
    a = np.clip((d * u).sum(axis=2) / (u * u).sum(), 0.0, 1.0)
    off = np.sqrt(((d - a[..., None] * u) ** 2).sum(axis=2))
    return (a > 0.5) & (off < 45.0)


def runs_of(mask_1d, min_len=1):
    idx = np.nonzero(mask_1d)[0]
    if len(idx) == 0:
        return []
    parts = np.split(idx, np.nonzero(np.diff(idx) > 1)[0] + 1)
    return [(int(p[0]), int(p[-1])) for p in parts if len(p) >= min_len]


def spines(rgb):
    """(top, bottom, left, right): the bottom spine's row, the left spine's column and their far ends - grey (axes styled
    by nice_plotting.format_axes) or black"""
    r, g, b = rgb[..., 0], rgb[..., 1], rgb[..., 2]
    ink = (np.abs(r - g) < 8) & (np.abs(g - b) < 8) & (r < 200)
    per_col, per_row = ink.sum(axis=0), ink.sum(axis=1)
    left = int(np.nonzero(per_col >= 0.95 * per_col.max())[0].min())          # (a full frame: the left / bottom one)
    bottom = int(np.nonzero(per_row >= 0.95 * per_row.max())[0].max())
    cols, rows = np.nonzero(ink[bottom])[0], np.nonzero(ink[:, left])[0]
    cols = cols[cols >= left]
    return int(rows.min()), bottom, left, int(cols.max())


def legend_frame(rgb, box, min_run=25):
    """bounding box (row0, row1, col0, col1) of the legend's light-grey frame: long horizontal and vertical runs of the
    frame colour (the anti-aliased rims of black dashes are a few pixels long)"""
    top, bot, left, right = box
    r, g, b = rgb[..., 0], rgb[..., 1], rgb[..., 2]
    grey = (np.abs(r - g) < 6) & (np.abs(g - b) < 6) & (r > 190) & (r < 240)
    ys, xs = [], []
    for y in range(top, bot):
        for a, b_ in runs_of(grey[y, left + 2:right + 1], min_run):
            ys += [y]
            xs += [a + left + 2, b_ + left + 2]
    for x in range(left + 2, right + 1):
        for a, b_ in runs_of(grey[top:bot, x], min_run):
            xs += [x]
            ys += [a + top, b_ + top]
    assert ys, "no legend frame"
    return min(ys) - 2, max(ys) + 2, min(xs) - 3, max(xs) + 3


def black_frames(rgb, box, min_len=30):
    """bounding boxes of black rectangles drawn inside the axes (an inset's frame): long black horizontal runs paired
    top / bottom over the same columns"""
    top, bot, left, right = box
    dark = rgb.max(axis=2) < 60.0
    found = []
    for y in range(top + 1, bot - 1):
        for a, b in runs_of(dark[y, left + 2:right + 1], min_len):
            found.append((y, a + left + 2, b + left + 2))
    frames = []
    for (y0, a0, b0) in found:
        for (y1, a1, b1) in found:
            if y1 > y0 + 8 and abs(a0 - a1) <= 1 and abs(b0 - b1) <= 1 and dark[y0:y1 + 1, a0].all() and dark[y0:y1 + 1, b0].all():
                frames.append((y0, y1, a0, b0))
    return frames


def wall_rows(rgb, box, skip, colour=BLACK, min_cover=0.2):
    """centre rows of the horizontal dash-dot lines drawn in black"""
    top, bot, left, right = box
    w = np.clip(1.0 - rgb.max(axis=2) / 255.0, 0.0, 1.0)
    w[w < 0.6] = 0.0
    for (r0, r1, c0, c1) in skip:
        w[max(r0, 0):r1 + 1, max(c0, 0):c1 + 1] = 0.0
    w[:, :left + 2] = 0.0
    w[:, right + 1:] = 0.0
    w[:top] = 0.0
    w[bot - 1:] = 0.0
    prof = (w > 0.5).sum(axis=1)
    strong = prof > min_cover * (right - left)
    out, y = [], 0
    while y < len(prof):
        if strong[y]:
            y1 = y
            while y1 + 1 < len(prof) and strong[y1 + 1]:
                y1 += 1
            lo, hi = max(0, y - 1), min(len(prof) - 1, y1 + 1)
            ww = w[lo:hi + 1].sum(axis=1)
            out.append(float((ww * (np.arange(lo, hi + 1) + 0.5)).sum() / ww.sum()))
            y = y1 + 1
        else:
            y += 1
    return out


def dot_centroids(rgb, box, skip):
    """centroids (x, y) of the isolated dots of the dotted black trajectory: small dark blobs with no coloured pixel near"""
    from scipy import ndimage
    top, bot, left, right = box
    dark = np.clip(1.0 - rgb.max(axis=2) / 255.0, 0.0, 1.0)
    dark[saturated(rgb)] = 0.0
    mask = dark > 0.15
    mask[:top] = False
    mask[bot - 1:] = False
    mask[:, :left + 2] = False
    mask[:, right + 1:] = False
    for (r0, r1, c0, c1) in skip:
        mask[max(r0, 0):r1 + 1, max(c0, 0):c1 + 1] = False
    lab, n = ndimage.label(mask, structure=np.ones((3, 3)))
    near_colour = ndimage.binary_dilation(saturated(rgb), iterations=2)
    out = []
    for k in range(1, n + 1):
        ys, xs = np.nonzero(lab == k)
        if len(ys) > 16 or ys.max() - ys.min() > 3 or xs.max() - xs.min() > 3 or near_colour[ys, xs].any():
            continue
        w = dark[ys, xs]
        if w.sum() >= 1.5:
            out.append(((w * (xs + 0.5)).sum() / w.sum(), (w * (ys + 0.5)).sum() / w.sum()))
    return np.array(out)


def dots_off_the_curve(points, cal, curve):
    """normal distances (pixels) of the dot centroids from the known curve drawn through calibration `cal`; dots further
    than 1.2 px are other artists' (a wall's dots)"""
    tt = np.linspace(0.0, T_END, 16001)
    cx, cy = cal.x(tt), cal.y(curve(tt))
    d = np.sqrt(((points[:, 0:1] - cx[None, :]) ** 2 + (points[:, 1:2] - cy[None, :]) ** 2).min(axis=1))
    return d[d < 1.2]


def column_bands(mask, box, skip_cols=(), skip_boxes=(), rows=None, cols=None):
    """[(col, first row, last row)] of every vertical run of `mask` inside the axes; whole columns touching a `skip_cols`
    range are left out (a legend or an inset lies OVER the curves there), `skip_boxes` only blank their pixels"""
    top, bot, left, right = box
    r0_, r1_ = rows if rows is not None else (top + 1, bot - 1)
    c0_, c1_ = cols if cols is not None else (left + 2, right)
    out = []
    for c in range(c0_, c1_ + 1):
        if any(a <= c <= b for a, b in skip_cols):
            continue
        col = mask[:, c].copy()
        col[:r0_] = False
        col[r1_ + 1:] = False
        for (r0, r1, a, b) in skip_boxes:
            if a <= c <= b:
                col[max(r0, 0):r1 + 1] = False
        out += [(c, a, b) for a, b in runs_of(col)]
    return out


def hidden_centre_intervals(bands, cal, width):
    """coloured bands a HIDDEN curve of line width `width` lies under -> the values its centre can take"""
    t, lo, hi = [], [], []
    for c, a, b in bands:
        y_top, y_bot = float(a), float(b + 1)
        if y_bot - y_top <= width:
            y_top = y_bot = 0.5 * (y_top + y_bot)
        else:
            y_top, y_bot = y_top + 0.5 * width, y_bot - 0.5 * width
        t.append(cal.t(c + 0.5))
        hi.append(cal.v(y_top))
        lo.append(cal.v(y_bot))
    return np.array(t), np.array(lo), np.array(hi)


def visible_centre_intervals(bands, cal, width):
    """pixels of a curve's OWN colour (possibly only what another curve leaves visible) -> the values its centre can
    take: every visible pixel lies within half a line width of the centre"""
    t, lo, hi = [], [], []
    for c, a, b in bands:
        y_top, y_bot = float(a), float(b + 1)
        c_lo, c_hi = y_bot - 0.5 * width, y_top + 0.5 * width          # (rows: c_lo <= centre <= c_hi)
        if c_lo > c_hi:
            c_lo = c_hi = 0.5 * (c_lo + c_hi)
        t.append(cal.t(c + 0.5))
        hi.append(cal.v(c_lo))
        lo.append(cal.v(c_hi))
    return np.array(t), np.array(lo), np.array(hi)


def inset_calibration(name, cal, edges, inset, frame):
    """mpl_toolkits' zoomed_inset_axes(ax, zoom, loc): a box of `zoom` times the parent's scale showing xlim x ylim, anchored
    inside the parent axes (9 = upper centre, 7 = centre right) half a font size (legend.fontsize 8 pt: nice_plotting.
    latexify) from its edge.  Its 1-px frame lines are drawn in the pixels round(edge) - asserted against the frame found
    in the figure - so the placement is known to a fraction of a pixel although the frame itself is snapped."""
    y_top, y_bot, x_lo, x_hi = edges
    zoom, xlim, ylim = inset["zoom"], inset["xlim"], inset["ylim"]
    w, h = zoom * (xlim[1] - xlim[0]) * cal.sx, zoom * (ylim[1] - ylim[0]) * cal.sy
    pad = 0.5 * 8.0 * 100.0 / 72.0
    if inset["loc"] == 9:
        left, top = 0.5 * (x_lo + x_hi) - 0.5 * w, y_top + pad
    else:
        assert inset["loc"] == 7
        left, top = x_hi - pad - w, 0.5 * (y_top + y_bot) - 0.5 * h
    f0, f1, g0, g1 = frame
    want = (np.floor(top + 0.5), np.floor(top + h + 0.5), np.floor(left + 0.5), np.floor(left + w + 0.5))
    assert max(abs(a - b) for a, b in zip(want, frame)) <= 1, (name, "inset frame", want, frame)
    exact = sum(a == b for a, b in zip(want, frame))
    ical = Calib(left - xlim[0] * cal.sx * zoom, cal.sx * zoom, top + h + ylim[0] * cal.sy * zoom, cal.sy * zoom)
    print("%-22s inset placed at columns %.2f ... %.2f, rows %.2f ... %.2f; frame drawn in %s (%d of 4 edges = round)"
          % (name, left, left + w, top, top + h, (g0, g1, f0, f1), exact))
    return ical


def put(out, key, tlh, cal, t_range=(0.0, T_END)):
    t, lo, hi = tlh
    keep = (t >= t_range[0]) & (t <= t_range[1])
    out[key + "_t"], out[key + "_lo"], out[key + "_hi"] = t[keep], lo[keep], hi[keep]
    out[key + "_pixel"] = cal.pixel
    return int(len(np.unique(t[keep])))


SMALL = (350.0, 0.7 * 2.1636 * 100.0)         # nice_plotting.latexify(fig_width=3.5, fig_height=0.7 * 2.1636) at 100 dpi
LARGE = (640.0, 480.0)                        # matplotlib's default figure


def check_layout(name, rgb, cal, edges, walls=None, trajectory=None, skip=()):
    """the cross-checks of a layout calibration (module docstring); returns the axes box in whole pixels"""
    y_top, y_bot, x_lo, x_hi = edges
    top, bot, left, right = spines(rgb)
    # spines: 1-px lines snapped to the pixel holding the axes edge
    assert abs(bot + 0.5 - y_bot) <= 0.75 and abs(left + 0.5 - x_lo) <= 0.75, (name, "spines", bot, y_bot, left, x_lo)
    box = (int(round(y_top)), bot, left, int(round(x_hi)))
    notes = []
    if walls is not None:
        found = wall_rows(rgb, box, skip)
        for value in walls:
            want = float(cal.y(value))
            miss = min(abs(f - want) for f in found)
            assert miss <= 0.62, (name, "wall", value, want, found)        # snapped by up to half a pixel
            notes.append("wall %+.2f %.2f px" % (value, miss))
    if trajectory is not None:
        pts = dot_centroids(rgb, box, skip)
        d = dots_off_the_curve(pts, cal, trajectory)
        assert len(d) >= 10 and np.sqrt((d ** 2).mean()) < 0.3, (name, "dotted trajectory", len(d), np.sqrt((d ** 2).mean()))
        notes.append("%d trajectory dots rms %.2f px" % (len(d), np.sqrt((d ** 2).mean())))
    print("%-22s layout calibration: pixel = %.4f s x %.5f;  %s" % (name, cal.pixel[0], cal.pixel[1], ";  ".join(notes)))
    return box


def position_figure(out, name, cell, axis, inset=None):
    """cells 21-23 / 25-27: one tool coordinate of the four controllers against the dash-dot walls and the dotted
    desired trajectory"""
    rgb = html_png(MOE, cell)
    traj = lambda t: path_des(t)[axis]                                           # noqa: E731
    tt = np.arange(10001) * 0.008
    lo = min(WALLS[axis, 0], traj(tt).min(), HOME_P[axis])
    hi = max(WALLS[axis, 1], traj(tt).max(), HOME_P[axis])
    cal, edges = layout_calibration(rgb.shape, SMALL, margins(0.0, T_END), margins(lo, hi))
    box0 = (int(round(edges[0])), int(round(edges[1])), int(round(edges[2])), int(round(edges[3])))
    legend = legend_frame(rgb, box0)
    frames = black_frames(rgb, box0)
    skip = [legend] + [(a - 2, b + 2, c - 2, d + 2) for a, b, c, d in frames]
    box = check_layout(name, rgb, cal, edges, walls=WALLS[axis], trajectory=traj, skip=skip)
    skip_cols = [(legend[2], legend[3])] + [(c - 2, d + 2) for a, b, c, d in frames]
    if inset is not None:
        # mark_inset: the grey rectangle around the zoomed region and its connectors cross the curves from there on
        skip_cols.append((int(cal.x(inset["xlim"][0])) - 2, box[3]))
    n_u = put(out, name + "_union", hidden_centre_intervals(column_bands(saturated(rgb), box, skip_cols), cal, LINE_W), cal)
    n_p = put(out, name + "_pinv", visible_centre_intervals(column_bands(colour_mask(rgb, C2), box, skip_cols), cal, LINE_W), cal)
    print("%-22s %d columns with coloured pixels, pinv's green visible in %d" % (name, n_u, n_p))
    if inset is not None:
        (f0, f1, g0, g1), = frames
        ical = inset_calibration(name, cal, edges, inset, (f0, f1, g0, g1))
        ibox = (f0, f1, g0, g1)
        rng = dict(rows=(f0 + 2, f1 - 2), cols=(g0 + 2, g1 - 2))
        # cell 26 draws the wall into the inset AFTER the curves: there a curve can also hide under the wall's black dashes
        cover = saturated(rgb) | (rgb.max(axis=2) < 110.0)
        sat = saturated(rgb)
        bands = [(c, a, b) for c, a, b in column_bands(cover, ibox, **rng) if sat[a:b + 1, c].any()]
        n_u = put(out, name + "_inset_union", hidden_centre_intervals(bands, ical, LINE_W), ical, inset["xlim"])
        n_p = put(out, name + "_inset_pinv", visible_centre_intervals(column_bands(colour_mask(rgb, C2), ibox, **rng), ical, LINE_W),
                  ical, inset["xlim"])
        print("%-22s inset: pixel = %.4f s x %.5f; %d columns, green visible in %d" % (name, ical.pixel[0], ical.pixel[1], n_u, n_p))


def tick_rows(rgb, box):
    """pixel rows (indices) of the tick marks left of the left spine.  Agg draws a tick mark whose coordinate is y into
    pixel row floor(y + 0.5) (checked on the x ticks of cell 13, whose positions t = 0, 10, ... 80 are known: the nine
    columns 103, 159, 215, 272, 328, 384, 441, 497, 553 are floor(x + 0.5) of the layout's 102.55, 158.91, ... 553.45 and
    NOT floor(x)), so a tick in row p says y in [p - 0.5, p + 0.5)"""
    top, bot, left, right = box
    dark = rgb.max(axis=2) < 200.0
    rows = []
    for y in range(top - 2, bot + 3):
        c, run = left - 1, 0
        while c >= 0 and dark[y, c]:
            run += 1
            c -= 1
        if run >= 2:
            rows.append(float(y))
    return rows


def error_figure(out, name, cell, size, inset=None, e_min_known=None):
    """cells 13-16: the tracking error norm of the four controllers.  The largest plotted value is the START error (a
    known input: |p(UR5_home) - path_des(0)|); the smallest one is an OUTPUT of the runs, so the lower view limit is the
    one free parameter, fitted to the rows of the y tick marks (at 0, 0.2, ... / 0, 0.25, ...: whole multiples of the
    step counted from the tick nearest the bottom = 0, since the error is a norm and the axes start below it)"""
    rgb = html_png(MOE, cell)
    e0 = float(np.linalg.norm(HOME_P - path_des(0.0)))
    _, edges = layout_calibration(rgb.shape, size, margins(0.0, T_END), (0.0, 1.0))
    box0 = tuple(int(round(e)) for e in edges)
    ticks = np.array(tick_rows(rgb, (box0[0], box0[1], spines(rgb)[2], box0[3])))
    step = inset["tick_step"] if inset else 0.2
    labels = step * np.arange(len(ticks))[::-1]                 # bottom tick = 0
    best = None
    fits = []
    for e_min in np.linspace(0.0, 0.05, 1001):
        cal, _ = layout_calibration(rgb.shape, size, margins(0.0, T_END), margins(e_min, e0))
        fits.append((np.abs(cal.y(labels) - ticks).max(), e_min))
    admissible = [e for miss, e in fits if miss <= 0.5]          # every tick within its pixel row
    assert admissible, (name, "tick rows", min(fits))
    e_min = 0.5 * (min(admissible) + max(admissible))
    out[name + "_min"] = np.array([min(admissible), max(admissible)])      # the smallest error ANY of the four runs reaches
    if e_min_known is not None:
        # the same e_sim arrays are drawn into the large figure two cells earlier, which resolves their minimum better
        assert min(admissible) - 1e-9 <= e_min_known <= max(admissible) + 1e-9, (name, e_min_known, admissible[0], admissible[-1])
        e_min = e_min_known
    cal, _ = layout_calibration(rgb.shape, size, margins(0.0, T_END), margins(e_min, e0))
    print("%-22s smallest plotted error fitted to the tick rows: %.4f (admissible %.4f ... %.4f = +-%.2f px at the bottom)"
          % (name, e_min, min(admissible), max(admissible), 0.5 * (max(admissible) - min(admissible)) * cal.sy))
    legend = legend_frame(rgb, box0)
    frames = black_frames(rgb, box0)
    box = check_layout(name, rgb, cal, edges, skip=[legend])
    skip_cols = []
    skip_boxes = [legend] + [(a - 2, b + 2, c - 2, d + 2) for a, b, c, d in frames]
    if inset is not None:
        # the marked rectangle and the connectors to the inset's corners (grey) are drawn over the curves
        (f0, f1, g0, g1), = frames
        skip_cols.append((min(int(cal.x(inset["xlim"][0])), g0) - 2, max(int(cal.x(inset["xlim"][1])), g1) + 3))
    n_u = put(out, name + "_union", hidden_centre_intervals(column_bands(saturated(rgb), box, skip_cols, skip_boxes), cal, LINE_W), cal)
    n_p = put(out, name + "_pinv", visible_centre_intervals(column_bands(colour_mask(rgb, C2), box, skip_cols, skip_boxes), cal, LINE_W), cal)
    print("%-22s %d columns with coloured pixels, pinv's green visible in %d" % (name, n_u, n_p))
    if inset is not None:
        (f0, f1, g0, g1), = frames
        ical = inset_calibration(name, cal, edges, inset, (f0, f1, g0, g1))
        ibox = (f0, f1, g0, g1)
        rng = dict(rows=(f0 + 2, f1 - 2), cols=(g0 + 2, g1 - 2))
        n_u = put(out, name + "_inset_union", hidden_centre_intervals(column_bands(saturated(rgb), ibox, **rng), ical, LINE_W),
                  ical, inset["xlim"])
        n_p = put(out, name + "_inset_pinv", visible_centre_intervals(column_bands(colour_mask(rgb, C2), ibox, **rng), ical, LINE_W),
                  ical, inset["xlim"])
        print("%-22s inset: pixel = %.4f s x %.5f; %d columns, green visible in %d" % (name, ical.pixel[0], ical.pixel[1], n_u, n_p))
    return e_min


def mode_levels(rgb, box, cal, colour, skip_boxes, other=None):
    """a step curve (mode_sim): per pixel column the level where the stroke is horizontal - a thin run of the colour that
    continues at the same rows in a neighbouring column - and the columns a chattering run FILLS between two levels.
    -> (t, level), (t, low level, high level)"""
    mask = colour_mask(rgb, colour)
    bands = {}
    for c, a, b in column_bands(mask, box, skip_boxes=skip_boxes):
        bands.setdefault(c, []).append((a, b))
    level_t, level_v, fill = [], [], []
    for c, runs in sorted(bands.items()):
        for a, b in runs:
            if b - a + 1 <= 3:
                neighbours = bands.get(c - 1, []) + bands.get(c + 1, [])
                if any(abs(a - a2) <= 0 and abs(b - b2) <= 0 for a2, b2 in neighbours):
                    v = float(cal.v(0.5 * (a + b + 1)))
                    if abs(v - round(v)) < 0.2:
                        level_t.append(float(cal.t(c + 0.5)))
                        level_v.append(v)
            else:
                lo, hi = float(cal.v(b + 1)), float(cal.v(a))
                # a filled block: this column AND both neighbours are covered over (almost) a whole level step
                full = [r for r in bands.get(c - 1, []) + bands.get(c + 1, []) if r[1] - r[0] + 1 >= 0.9 * (b - a + 1)]
                if hi - lo >= 0.9 and len(full) >= 2:
                    fill.append((float(cal.t(c + 0.5)), np.ceil(lo - 0.2), np.floor(hi + 0.2)))
    return (np.array(level_t), np.array(level_v)), np.array(fill).reshape(-1, 3)


def mode_figures(out):
    """cell 17: mode_sim of the two PseudoInverseControllers over xlim [-1, 50] ('separate': the three 1-D walls, 8
    modes, green dashed; 'multidim': blue, drawn over it); cell 18: 'multidim' alone over the 80 s.  The modes are whole
    numbers starting at 0 (mode_sim[0] = 0), the largest one drawn sets the upper view limit: read off as the topmost
    level, 5 (cell 17) and 1 (cell 18)."""
    rgb = html_png(MOE, 17)
    cal, edges = layout_calibration(rgb.shape, SMALL, (-1.0, 50.0), margins(0.0, 5.0))
    box0 = tuple(int(round(e)) for e in edges)
    legend = legend_frame(rgb, box0)
    box = check_layout("moe_modes", rgb, cal, edges, skip=[legend])
    for curve, colour in (("multidim", PURE_BLUE), ("separate", PURE_GREEN)):
        (t, v), fill = mode_levels(rgb, box, cal, colour, [legend])
        keep = t >= 0.0
        out["moe_modes_%s_t" % curve], out["moe_modes_%s_lo" % curve], out["moe_modes_%s_hi" % curve] = t[keep], v[keep], v[keep]
        out["moe_modes_%s_pixel" % curve] = cal.pixel
        print("moe_modes %-9s %d level samples, levels %s, %d filled columns" % (
            curve, keep.sum(), sorted(set(int(round(x)) for x in v[keep])), len(fill)))
        if len(fill):
            out["moe_modes_%s_fill" % curve] = fill
    rgb = html_png(MOE, 18)
    cal, edges = layout_calibration(rgb.shape, SMALL, margins(0.0, T_END), margins(0.0, 1.0))
    box = check_layout("moe_modes_full", rgb, cal, edges)
    (t, v), fill = mode_levels(rgb, box, cal, C0, [])
    keep = (t >= 0.0) & (t <= T_END)
    out["moe_modes_full_multidim_t"], out["moe_modes_full_multidim_lo"], out["moe_modes_full_multidim_hi"] = t[keep], v[keep], v[keep]
    out["moe_modes_full_multidim_pixel"] = cal.pixel
    print("moe_modes_full         %d level samples, levels %s" % (keep.sum(), sorted(set(int(round(x)) for x in v[keep]))))


def collect(out):
    for name, cell, axis in (("moe_x_singular", 21, 0), ("moe_y_singular", 22, 1), ("moe_z_singular", 23, 2),
                             ("moe_x_multidim", 25, 0), ("moe_z_multidim", 27, 2)):
        position_figure(out, name, cell, axis)
    # cell 26: zoomed_inset_axes(ax, 7, loc=7), xlim (58, 62), ylim (0.37, 0.41): the approach to the wall y_max = 0.4
    position_figure(out, "moe_y_multidim", 26, 1, inset=dict(zoom=7.0, loc=7, xlim=(58.0, 62.0), ylim=(0.37, 0.41)))
    e_min_singular = error_figure(out, "moe_e_singular", 13, LARGE)
    e_min_multidim = error_figure(out, "moe_e_multidim", 14, LARGE)
    # cells 15, 16: the same curves small, with zoomed_inset_axes(ax, 6, loc=9), xlim (41, 45), ylim (0.05, 0.13)
    ins = dict(zoom=6.0, loc=9, xlim=(41.0, 45.0), ylim=(0.05, 0.13), tick_step=0.25)
    error_figure(out, "moe_e_singular_small", 15, SMALL, inset=ins, e_min_known=e_min_singular)
    error_figure(out, "moe_e_multidim_small", 16, SMALL, inset=ins, e_min_known=e_min_multidim)
    mode_figures(out)


# ---- ur5_dual_quaternion_comparison_of_controllers.ipynb: error norms on LOG axes (cells 19, 20, 41, 42) ------------------
DQC = "ur5_dual_quaternion_comparison_of_controllers.ipynb"
DQ_T_END = 45.0                      # cell 17 / 39: 4500 ticks of 0.01 s


def major_tick_rows(rgb, box, min_len=4):
    """pixel rows of the MAJOR tick marks left of the left spine (3.5 pt = 5 px long; the minor ticks of a log axis are
    2 pt = 3 px)"""
    top, bot, left, right = box
    dark = rgb.max(axis=2) < 200.0
    rows = []
    for y in range(top - 2, bot + 3):
        c, run = left - 1, 0
        while c >= 0 and dark[y, c]:
            run += 1
            c -= 1
        if run >= min_len:
            rows.append(float(y))
    return rows


def log_error_figure(out, name, cell, top_decade, per_tick, hi_known=None):
    """the error norm of the four controllers (drawn in the order qp, pinv, mpc, nlp - cell 17's printed order) on a log
    axis: values are stored as log10(error).  The view limits are the data's range in log10 widened by 5 %; both ends are
    OUTPUTS of the runs (the start value, where it is the largest, is an input and is cross-checked), so they are fitted to
    the rows of the major tick marks - whole decades, the topmost one 10^`top_decade`, `per_tick` decades apart (both read
    off the stored figure's labels by eye) - under the rule that a tick at coordinate y sits in pixel row floor(y + 0.5)."""
    rgb = html_png(DQC, cell)
    _, edges = layout_calibration(rgb.shape, LARGE, margins(0.0, DQ_T_END), (0.0, 1.0))
    box0 = tuple(int(round(e)) for e in edges)
    sp = spines(rgb)
    ticks = np.array(major_tick_rows(rgb, (box0[0], box0[1], sp[2], box0[3])))
    assert len(ticks) >= 2 and np.abs(np.diff(ticks) - np.diff(ticks).mean()).max() <= 1.0, (name, "major ticks", ticks)
    labels = top_decade - per_tick * np.arange(len(ticks))
    # rows are affine in the label: least squares for (row of decade 0, rows per decade), then the view limits
    A = np.stack([np.ones(len(ticks)), -labels], axis=1)
    (r0, rpd), _, _, _ = np.linalg.lstsq(A, ticks, rcond=None)
    y_top, y_bot = edges[0], edges[1]
    v_hi, v_lo = (r0 - y_top) / rpd, (r0 - y_bot) / rpd          # log10 at the axes' top / bottom edge
    lo, hi = v_lo + (v_hi - v_lo) * 0.05 / 1.1, v_hi - (v_hi - v_lo) * 0.05 / 1.1
    miss = np.abs(A.dot([r0, rpd]) - ticks).max()
    if hi_known is not None:
        miss = max(miss, abs(hi - hi_known) * rpd)
    best = (miss, per_tick, lo, hi, rpd)
    miss, per_tick, lo, hi, rpd = best
    assert miss <= 0.75, (name, "tick rows / known top value", miss)
    cal, edges = layout_calibration(rgb.shape, LARGE, margins(0.0, DQ_T_END), margins(lo, hi))
    print("%-22s log axis: %d major ticks, %d decade(s) apart, data range 1e%.3f ... 1e%.3f (fit within %.2f px%s); pixel = "
          "%.4f s x %.4f decades" % (name, len(ticks), per_tick, lo, hi, miss,
                                    "" if hi_known is None else ", known top 1e%.4f" % hi_known, cal.pixel[0], cal.pixel[1]))
    legend = legend_frame(rgb, box0)
    box = check_layout(name, rgb, cal, edges, skip=[legend])
    skip_boxes = [legend]
    n_u = put(out, name + "_union", hidden_centre_intervals(column_bands(saturated(rgb), box, (), skip_boxes), cal, LINE_W), cal,
              (0.0, DQ_T_END))
    for curve, colour in (("qp", C0), ("pinv", C1)):
        n_c = put(out, name + "_" + curve, visible_centre_intervals(column_bands(colour_mask(rgb, colour), box, (), skip_boxes),
                                                                   cal, LINE_W), cal, (0.0, DQ_T_END))
        print("%-22s %s visible in %d of %d columns" % (name, curve, n_c, n_u))
    out[name + "_range"] = np.array([lo, hi])
    if name == "dqc_cart_dist":
        # the pinv run CHATTERS from t = 8 s on (its target is out of reach): the stored curve is a band.  Its extent -
        # the values the curve's centre sweeps, half a line width inside the band's pixels - over the columns left of
        # the legend:
        m = colour_mask(rgb, C1)
        tops, bots = [], []
        for c in range(int(cal.x(12.0)), min(int(cal.x(38.0)), legend[2] - 2)):
            rows = np.nonzero(m[box[0]:box[1], c])[0] + box[0]
            tops.append(rows.min())
            bots.append(rows.max() + 1)
        assert max(tops) - min(tops) <= 1 and max(bots) - min(bots) <= 1, (name, "band not flat", set(tops), set(bots))
        out[name + "_pinv_band"] = np.array([cal.v(np.median(bots) - 0.5 * LINE_W), cal.v(np.median(tops) + 0.5 * LINE_W)])
        print("%-22s pinv chatter band (centre of the line): %.5f ... %.5f" % (name, *(10.0 ** out[name + "_pinv_band"])))


def collect_dq(out):
    e0 = {"cart": np.log10(1.0192018)}          # |p_tool0| at UR5_home, the KAT the notebook prints (cell 7)
    log_error_figure(out, "dqc_cart_dist", 19, 0, 1, hi_known=e0["cart"])
    log_error_figure(out, "dqc_quat_dist", 20, 0, 2)
    log_error_figure(out, "dqc_Q_dist1", 41, -1, 2)
    log_error_figure(out, "dqc_Q_dist2", 42, 0, 2)


if __name__ == "__main__":
    arrays = {}
    collect(arrays)
    collect_dq(arrays)
    print(len(arrays), "arrays")
