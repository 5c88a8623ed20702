"""The 3-D figures the reference's notebooks store (`common_plots.frame_3d`: the tool's path in black and the tips of its
frame's axes, p + 0.1 R e_i, as red / green / blue dashed curves, on a default `Axes3D`) as pins of the POSE trajectory.

What is stored (tests/golden/notebook_figures.npz, written by make_figure_pins.py): for every figure the coordinates of the
pixels of each curve colour inside the axes box - data read off the figure, nothing else.  What turns a simulated run
into pixels is matplotlib's (2.x) own projection, restated here from its documentation and checked on the figures'
dots (the desired frame's origin and axis tips: known points, no simulation involved):

  view limits   the data range of everything drawn (the four curves and the dots) widened by 5 % on each side
                (axes.xmargin / ymargin / zmargin = 0.05, autolimit_mode "data")
  world -> unit cube -> eye at elev 30, azim -60, distance 10 from the cube's centre, up = z -> perspective (front / back
  at -+ distance) -> the 2-D axes limits (-0.95 / dist, 0.9 / dist) in both directions, axes filling the whole
  640 x 480 canvas (`Axes3D(plt.figure())`), rows counted from the top.

Because the view limits come from the simulated curves themselves, a run that leaves the stored one anywhere near its
extremes moves EVERY pixel."""
import numpy as np

COLOURS = ("k", "r", "g", "b")


class OldAxes3D(object):
    """matplotlib 2.x `Axes3D.get_proj` for given view limits [(xmin, xmax), (ymin, ymax), (zmin, zmax)]"""

    def __init__(self, limits, elev=30.0, azim=-60.0, dist=10.0, width=640, height=480):
        (x0, x1), (y0, y1), (z0, z1) = limits
        dx, dy, dz = x1 - x0, y1 - y0, z1 - z0
        world = np.array([[1 / dx, 0, 0, -x0 / dx], [0, 1 / dy, 0, -y0 / dy], [0, 0, 1 / dz, -z0 / dz], [0, 0, 0, 1.0]])
        relev, razim = np.pi * elev / 180, np.pi * azim / 180
        centre = np.array([0.5, 0.5, 0.5])
        eye = centre + dist * np.array([np.cos(razim) * np.cos(relev), np.sin(razim) * np.cos(relev), np.sin(relev)])
        n = (eye - centre) / np.linalg.norm(eye - centre)
        u = np.cross([0.0, 0.0, 1.0], n)
        u /= np.linalg.norm(u)
        v = np.cross(n, u)
        rot, shift = np.eye(4), np.eye(4)
        rot[0, :3], rot[1, :3], rot[2, :3] = u, v, n
        shift[:3, 3] = -eye
        zf, zb = -dist, dist
        persp = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, (zf + zb) / (zf - zb), -2 * zf * zb / (zf - zb)], [0, 0, -1, 0.0]])
        self.M = persp.dot(rot.dot(shift)).dot(world)
        self.lo, self.hi = -0.95 / dist, 0.9 / dist
        self.width, self.height = width, height
        self.limits = limits

    def pixels(self, X):
        """[n, 3] world points -> [n, 2] (column, row) on the canvas"""
        X = np.atleast_2d(np.asarray(X, dtype=float))
        h = np.c_[X, np.ones(len(X))].dot(self.M.T)
        x, y = h[:, 0] / h[:, 3], h[:, 1] / h[:, 3]
        span = self.hi - self.lo
        return np.c_[(x - self.lo) / span * self.width, self.height - (y - self.lo) / span * self.height]

    def box_corners(self):
        (x0, x1), (y0, y1), (z0, z1) = self.limits
        return self.pixels([[x, y, z] for x in (x0, x1) for y in (y0, y1) for z in (z0, z1)])


def autoscaled_limits(points, margin=0.05):
    """view limits of everything drawn: [n, 3] points"""
    lo, hi = points.min(axis=0), points.max(axis=0)
    d = (hi - lo) * margin
    return [(lo[a] - d[a], hi[a] + d[a]) for a in range(3)]


def frame_curves(p_sim, R_sim):
    """what frame_3d draws: {"k": path, "r" / "g" / "b": p + 0.1 R e_i}"""
    out = {"k": np.asarray(p_sim)}
    for i, c in enumerate("rgb"):
        out[c] = p_sim + 0.1 * R_sim[:, :, i]
    return out


def frame_dots(T_des=None, p_des=None):
    if T_des is not None:
        T_des = np.asarray(T_des, dtype=float)
        return {"k": T_des[:3, 3]} | {c: T_des[:3, 3] + 0.1 * T_des[:3, i] for i, c in enumerate("rgb")}
    return {"k": np.asarray(p_des, dtype=float)}


def axes_of(p_sim, R_sim, T_des=None, p_des=None, limits=None, **kw):
    curves, dots = frame_curves(p_sim, R_sim), frame_dots(T_des, p_des)
    if limits is None:
        limits = autoscaled_limits(np.vstack(list(curves.values()) + [np.atleast_2d(d) for d in dots.values()]))
    return OldAxes3D(limits, **kw), curves, dots


# ---- reading a stored figure ------------------------------------------------------------------------------------------
def box_mask(rgb, shrink=6):
    """the axes box on the canvas: its panes are light grey on a white figure; eroded by `shrink` pixels so that the
    black axis lines and tick marks on its outline stay out"""
    from scipy import ndimage
    v = rgb[..., :3]
    v = v / 255.0 if v.max() > 1.5 else v
    shaded = (v.min(axis=2) < 0.985) & (v.min(axis=2) > 0.93) & (v.max(axis=2) - v.min(axis=2) < 0.02)   # pane grey only:
    # (the title's letters above the box stay apart: their fringes are specks an opening removes; the grid lines that
    # cut the panes into cells are bridged by the closing)
    cells = ndimage.binary_closing(ndimage.binary_opening(shaded, iterations=1), iterations=3)
    lab, n = ndimage.label(cells)
    if n == 0:
        return np.zeros(shaded.shape, bool)
    sizes = ndimage.sum(cells, lab, index=np.arange(1, n + 1))
    box = ndimage.binary_fill_holes(lab == (1 + int(np.argmax(sizes))))
    return ndimage.binary_erosion(box, iterations=shrink)


def colour_masks(rgb):
    """pixels that are mostly one of matplotlib's "k" / "r" / "g" / "b" ((0, 0.5, 0) for green) over the light panes"""
    v = rgb[..., :3]
    v = v / 255.0 if v.max() > 1.5 else v
    r, g, b = v[..., 0], v[..., 1], v[..., 2]
    return {"k": (v.max(axis=2) < 0.35),
            "r": (r > 0.7) & (g < 0.45) & (b < 0.45),
            "g": (g > 0.35) & (g < 0.8) & (r < 0.45 * g / 0.5) & (b < 0.45 * g / 0.5) & (g - np.maximum(r, b) > 0.25),
            "b": (b > 0.7) & (r < 0.45) & (g < 0.45)}


def digitise(rgb):
    """{colour: [n, 2] (column, row) centres of that colour's pixels inside the axes box}"""
    box = box_mask(rgb)
    out = {}
    for c, mask in colour_masks(rgb).items():
        rows, cols = np.nonzero(mask & box)
        out[c] = np.c_[cols + 0.5, rows + 0.5]
    return out


# ---- comparing -------------------------------------------------------------------------------------------------------
def dense(polyline, step=0.4):
    """the projected curve sampled every `step` pixels of its length (a tool at rest adds no samples)"""
    seg = np.linalg.norm(np.diff(polyline, axis=0), axis=1)
    s = np.concatenate([[0.0], np.cumsum(seg)])
    if s[-1] < step:
        return polyline[:1].copy()
    at = np.arange(0.0, s[-1], step)
    return np.c_[np.interp(at, s, polyline[:, 0]), np.interp(at, s, polyline[:, 1])]


def deviations(stored, axes, curves, dots, dot_radius=6.0):
    """per colour: (worst distance of a stored pixel from the simulated curve, share of the simulated curve that has a
    stored pixel of its colour - or of a curve drawn over it - within 1.5 pixels, pixels compared)"""
    from scipy.spatial import cKDTree
    drawn = {c: dense(axes.pixels(X)) for c, X in curves.items()}
    dot_px = {c: axes.pixels(d)[0] for c, d in dots.items()}
    everything = cKDTree(np.vstack([stored[c] for c in COLOURS if len(stored[c])]))
    out = {}
    for c in COLOURS:
        px = stored[c]
        if c in dot_px and len(px):
            px = px[np.linalg.norm(px - dot_px[c], axis=1) > dot_radius]      # (the dot itself is not the curve)
        if len(px) == 0:
            out[c] = (0.0, 1.0, 0)
            continue
        d_to_curve = cKDTree(drawn[c]).query(px)[0]
        covered = everything.query(drawn[c])[0] < 1.5
        out[c] = (float(d_to_curve.max()), float(covered.mean()), int(len(px)))
    return out


# ---- the figures ------------------------------------------------------------------------------------------------------
DQC_NOTEBOOK = "ur5_dual_quaternion_comparison_of_controllers.ipynb"
# (constraint, controller) -> cell.  PINV(Q_dist2), cell 50, is read as well but is NOT a pin: that run starts in the
# home singularity of the 8-row task and its first ticks ask for hundreds of rad/s, clipped to pi / 5 - a start moved
# by 1e-9 rad changes those velocities by 2 %, one moved by 1e-6 ends with joint 6 at -1.88 instead of -1.39 rad; the
# stored run (tool displaced 8 mm) and the oracle's (0.2 mm) are two members of that family.  Its ERROR NORM, which the
# distance to the target dominates, is pinned all the same (dqc_Q_dist2_pinv).
DQC_FRAMES = {("cart_dist", "qp"): 22, ("cart_dist", "pinv"): 23, ("quat_dist", "qp"): 26, ("quat_dist", "pinv"): 27,
              ("Q_dist1", "qp"): 44, ("Q_dist1", "pinv"): 46, ("Q_dist2", "qp"): 48, ("Q_dist2", "pinv"): 50}
DQC_P_DES = [0.5, 0.5, 0.5]                                     # cell 12
DQC_T_DES = [[1.0, 0, 0, 0.2], [0, 1.0, 0, 0.2], [0, 0, 1.0, 0.75], [0, 0, 0, 1.0]]      # cell 33: T_rpy([0.2, 0.2, 0.75], 0, 0, 0)


def dqc_target(which):
    return dict(T_des=np.array(DQC_T_DES)) if which.startswith("Q_") else dict(p_des=np.array(DQC_P_DES))


# ur5_dual_quaternion_vs_transformation_matrix.ipynb cells 33, 34: the ReactiveQPController's runs on Q_dist2 and T_dist2
# (1000 ticks of 0.008 s from UR5_home to a frame rolled by 5 degrees at (0.5, 0, 0.5)); `nice_plotting.latexify(3.5)`
# makes the canvas 350 x 216, the cells set the view limits themselves.  (Cells 35, 36 - the PseudoInverseController with
# damping 1e-26 from the singular home - are rounding noise / 1e-26 at the start)
DQTM_NOTEBOOK = "ur5_dual_quaternion_vs_transformation_matrix.ipynb"
DQTM_FRAMES = {"Q_dist2": 33, "T_dist2": 34}
DQTM_LIMITS = [(-0.5, 1.0), (-0.5, 1.0), (0.0, 1.0)]
DQTM_CANVAS = (350, 216)


def dqtm_target():
    roll = 5.0 * np.pi / 180.0                                   # cell 16: rotation_rpy(roll, 0, 0) at (0.5, 0, 0.5)
    T = np.eye(4)
    T[:3, :3] = [[1.0, 0.0, 0.0], [0.0, np.cos(roll), -np.sin(roll)], [0.0, np.sin(roll), np.cos(roll)]]
    T[:3, 3] = [0.5, 0.0, 0.5]
    return T


# ur5_transformation_matrix_comparison_of_controllers.ipynb cells 17 (ReactiveQPController, the only figure its point run
# has) and 33 (PseudoInverseController, the run of cell 32): inline backend, a 6 x 4 inch figure at 72 dpi = 432 x 288
# canvas pixels saved with bbox_inches="tight", which pads 7.2 px and follows the tick labels beyond the canvas: the
# stored 453 x 309 image sits at an offset only the figure can tell.  It is read off the black dot at p_des = (0.5, 0.5,
# 0.5) (stored: the dot's centroid); view limits (-1, 1)^3 as the cells set them; one pixel = 6 mm.
TM_NOTEBOOK = "ur5_transformation_matrix_comparison_of_controllers.ipynb"
TM_FRAMES = {"qp": 17, "pinv": 33}
TM_LIMITS = [(-1.0, 1.0), (-1.0, 1.0), (-1.0, 1.0)]
TM_CANVAS = (432, 288)
TM_P_DES = [0.5, 0.5, 0.5]


def black_dot(rgb):
    """centroid (column, row) of the round black marker inside the axes box: the compact blob among the black pixels (the
    path is a 1.5-pixel line, the dot a disc of 4.5)"""
    from scipy import ndimage
    mask = colour_masks(rgb)["k"] & box_mask(rgb)
    density = ndimage.uniform_filter(mask.astype(float), size=5)                   # a thin line fills 5 x 5 far less
    r0, c0 = np.unravel_index(int(np.argmax(density)), density.shape)
    assert density[r0, c0] * 25 >= 8, "no dot"
    rows, cols = np.nonzero(mask)
    near = (rows - r0) ** 2 + (cols - c0) ** 2 <= 3.5 ** 2
    return np.array([cols[near].mean() + 0.5, rows[near].mean() + 0.5])


def collect_frames(out, html_png, stored_png=None):
    """pixel lists of the frame_3d figures -> out["f3d_<constraint>_<controller>_<colour>"] ([n, 2] int16, column and row
    of each pixel)"""
    if stored_png is not None:
        for kind, cell in TM_FRAMES.items():
            rgb = stored_png(TM_NOTEBOOK, cell)
            assert rgb.shape[:2] == (309, 453)
            for c, px in digitise(rgb).items():
                out["f3d_tm_point_%s_%s" % (kind, c)] = np.floor(px).astype(np.int16)
            out["f3d_tm_point_%s_dot" % kind] = black_dot(rgb)
    for (which, kind), cell in DQC_FRAMES.items():
        rgb = html_png(DQC_NOTEBOOK, cell)
        assert rgb.shape[:2] == (480, 640)
        for c, px in digitise(rgb).items():
            out["f3d_%s_%s_%s" % (which, kind, c)] = np.floor(px).astype(np.int16)
    for which, cell in DQTM_FRAMES.items():
        rgb = html_png(DQTM_NOTEBOOK, cell)
        assert rgb.shape[:2] == DQTM_CANVAS[::-1]
        for c, px in digitise(rgb).items():
            out["f3d_dqtm_%s_qp_%s" % (which, c)] = np.floor(px).astype(np.int16)


def stored_frames(figs, which, kind, prefix="f3d_"):
    return {c: figs["%s%s_%s_%s" % (prefix, which, kind, c)].astype(float) + 0.5 for c in COLOURS}


# ---- a 2-D figure read the same way: pixels of each curve colour inside an axes frame --------------------------------
# ur5_transformation_matrix_comparison_of_controllers.ipynb cell 31 (`common_plots.joints(sim_point_pinv)`): the six joint
# positions (top axes) and the six clamped joint speeds (bottom axes) of the PseudoInverseController's point run, in
# matplotlib's default colour cycle - the JOINT-space side of the run whose tool position cell 32 shows: which joints move
# is the pseudo-inverse's choice.  Inline backend, 72 dpi; the axes frames are found in the image (their spines), the view
# limits are the simulated data's range widened by 5 %; the legend covers the right end of the top axes (left out).
TAB10 = [(31, 119, 180), (255, 127, 14), (44, 160, 44), (214, 39, 40), (148, 103, 189), (140, 86, 75)]
TM_JOINTS_CELL = 31
TM_JOINTS_LEGEND_FROM = 328          # first pixel column of the legend over the top axes


def palette_masks(rgb, palette=TAB10, coverage=0.6, off=22.0):
    """per palette colour the pixels that are a mix of white and that colour with at least `coverage` of the colour"""
    v = rgb[..., :3].astype(float)
    v = v * 255.0 if v.max() <= 1.5 else v
    d = 255.0 - v
    best_off = np.full(v.shape[:2], np.inf)
    best = np.full(v.shape[:2], -1)
    cov = np.zeros(v.shape[:2])
    for k, c in enumerate(palette):
        u = 255.0 - np.array(c, dtype=float)
        a = np.clip((d * u).sum(axis=2) / (u * u).sum(), 0.0, 1.2)
        o = np.sqrt(((d - a[..., None] * u) ** 2).sum(axis=2))
        take = o < best_off
        best_off[take], best[take], cov[take] = o[take], k, a[take]
    return [(best == k) & (best_off < off) & (cov > coverage) for k in range(len(palette))]


def frames_2d(rgb):
    """[(top, bottom, left, right)] centres of the axes' spines, top axes first"""
    v = rgb[..., :3].astype(float)
    v = v * 255.0 if v.max() <= 1.5 else v
    dark = v.max(axis=2) < 90.0
    colsum = dark.sum(axis=0)
    cols = np.nonzero(colsum > 0.6 * colsum.max())[0]
    left, right = int(cols.min()), int(cols.max())
    width = right - left + 1
    rows = [y for y in range(dark.shape[0]) if dark[y, left:right + 1].sum() > 0.97 * width]
    assert len(rows) % 2 == 0 and len(rows) >= 2, rows
    return [(rows[2 * k] + 0.5, rows[2 * k + 1] + 0.5, left + 0.5, right + 0.5) for k in range(len(rows) // 2)]


def collect_joint_figure(out, stored_png):
    rgb = stored_png(TM_NOTEBOOK, TM_JOINTS_CELL)
    frames = frames_2d(rgb)
    assert len(frames) == 2
    from scipy import ndimage
    masks = palette_masks(rgb)
    # (where curves of different colours run together their anti-aliased rims mix into other palette colours - red over
    # green reads as brown: a pixel next to another colour's pixel is left out)
    clean = []
    for k, mask in enumerate(masks):
        others = np.zeros_like(mask)
        for j, other in enumerate(masks):
            if j != k:
                others |= other
        clean.append(mask & ~ndimage.binary_dilation(others, structure=np.ones((3, 3), bool)))
    masks = clean
    out["j2d_tm_pinv_frames"] = np.array(frames)
    for a, (top, bot, left, right) in enumerate(frames):
        for k, mask in enumerate(masks):
            m = mask.copy()
            m[:int(top) + 2] = False
            m[int(bot) - 1:] = False
            m[:, :int(left) + 2] = False
            m[:, (TM_JOINTS_LEGEND_FROM if a == 0 else int(right) - 1):] = False
            rows, cols = np.nonzero(m)
            out["j2d_tm_pinv_%s_%d" % ("q" if a == 0 else "dq", k)] = np.c_[cols, rows].astype(np.int16)


def deviations_2d(stored, frame, t, values, last_column=None):
    """`stored`: [colour] -> [n, 2] pixel centres; `values` [n_t, n_curves] drawn against `t` in an axes whose spines sit at
    `frame`, view limits = the data's range widened by 5 %.  Per curve (worst distance of a stored pixel from the drawn
    polyline, share of the polyline under ink, pixels)"""
    from scipy.spatial import cKDTree
    top, bot, left, right = frame
    t0, t1 = t.min(), t.max()
    v0, v1 = values.min(), values.max()
    t0, t1 = t0 - 0.05 * (t1 - t0), t1 + 0.05 * (t1 - t0)
    v0, v1 = v0 - 0.05 * (v1 - v0), v1 + 0.05 * (v1 - v0)
    col = left + (t - t0) / (t1 - t0) * (right - left)
    ink = cKDTree(np.vstack([s for s in stored if len(s)]))
    out = []
    for k in range(values.shape[1]):
        row = top + (v1 - values[:, k]) / (v1 - v0) * (bot - top)
        line = dense(np.c_[col, row])
        if last_column is not None:
            line = line[line[:, 0] < last_column - 2.0]
        px = stored[k]
        if len(px) == 0:
            out.append((0.0, float((ink.query(line)[0] < 1.5).mean()), 0))
            continue
        out.append((float(cKDTree(dense(np.c_[col, row])).query(px)[0].max()), float((ink.query(line)[0] < 1.5).mean()), len(px)))
    return out
