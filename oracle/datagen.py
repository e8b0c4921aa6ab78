"""Oracle: the synthetic-shape data generator on the CPU (TEST INFRASTRUCTURE).

Two parts:
 * OpenCV's scan-conversion rules for the three cv2 drawing calls the reference makes, restated in numpy from the algorithms of
   OpenCV's drawing.cpp (`cv_circle_masks`, `cv_poly_masks`, `cv_line`, `cv_clip_line`); `Cv2Stub` serves them to the reference's
   own generator when golden g14 is made (cv2 itself is not installed);
 * a restatement of train_val_data_generator.py:31-130 (scene -> images, all-in-focus, boundary / depth maps,
   city-block distances, Sobel maps), :165-182 (noise model, deterministic part) and :226-252 (patch crop + in-patch
   distance transform) with the same scipy.ndimage calls, pinned against the reference by golden g14.
"""
import numpy as np
from scipy.ndimage import binary_dilation, convolve, generate_binary_structure

CIRCLE, RECT, TRI = 0, 1, 2


# ---------------------------------------------------------------------------------------------------------------------------
# OpenCV's scan-conversion rules, restated (round 5).  cv2 is not installed here and OpenCV documents no pixel-exact rule, so
# these follow the ALGORITHMS of OpenCV 4.x `modules/imgproc/src/drawing.cpp`, which the three calls of
# train_val_data_generator.py:58-76 reach with their default arguments (lineType = LINE_8, shift = 0):
#   cv2.circle(img, c, r, color, -1 | 1)            -> circle() -> Circle(img, c, r, color, fill)      [thickness <= 1, LINE_8, shift 0]
#   cv2.drawContours(img, [pts], 0, color, 1)       -> a closed polyline: ThickLine(thickness 1) -> Line() per edge
#   cv2.drawContours(img, [pts], 0, color, -1)      -> CollectPolyEdges() (draws every edge with Line(), collects the non-horizontal
#                                                      edges in 16.16 fixed point) + FillEdgeCollection() (scan-line fill)
#   Line()                                          -> LineIterator(img, p1, p2, 8, leftToRight = true): clipLine() to the image
#                                                      rectangle, then the 8-connected Bresenham walk
# Status: a restatement from the published source as the builder knows it - not run against cv2 (absent offline).  Known answers
# worked by hand from those algorithms are held in tests/test_host_cpu.py.
XY_SHIFT = 16
XY_ONE = 1 << XY_SHIFT


def _trunc_div(a, b):
    """C++ integer division (truncation toward zero) of Python ints."""
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


def cv_clip_line(W, H, x1, y1, x2, y2):
    """clipLine(Size(W, H), pt1, pt2): -> (visible, x1, y1, x2, y2).  The products are formed in double and truncated, as OpenCV 4.x
    does; for image-sized coordinates that equals the integer division of OpenCV 3.x."""
    right, bottom = W - 1, H - 1
    if W <= 0 or H <= 0:
        return False, x1, y1, x2, y2
    c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8
    c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8
    if (c1 & c2) == 0 and (c1 | c2) != 0:
        if c1 & 12:
            a = 0 if c1 < 8 else bottom
            x1 += int(float(a - y1) * (x2 - x1) / (y2 - y1))
            y1 = a
            c1 = (x1 < 0) + (x1 > right) * 2
        if c2 & 12:
            a = 0 if c2 < 8 else bottom
            x2 += int(float(a - y2) * (x2 - x1) / (y2 - y1))
            y2 = a
            c2 = (x2 < 0) + (x2 > right) * 2
        if (c1 & c2) == 0 and (c1 | c2) != 0:
            if c1:
                a = 0 if c1 == 1 else right
                y1 += int(float(a - x1) * (y2 - y1) / (x2 - x1))
                x1 = a
                c1 = 0
            if c2:
                a = 0 if c2 == 1 else right
                y2 += int(float(a - x2) * (y2 - y1) / (x2 - x1))
                x2 = a
                c2 = 0
    return (c1 | c2) == 0, x1, y1, x2, y2


def cv_line(img, p1, p2):
    """Line(img, p1, p2, color, 8) on a boolean image: clip, then LineIterator's 8-connected walk from the left end point."""
    H, W = img.shape
    x1, y1, x2, y2 = int(p1[0]), int(p1[1]), int(p2[0]), int(p2[1])
    if not (0 <= x1 < W and 0 <= x2 < W and 0 <= y1 < H and 0 <= y2 < H):
        ok, x1, y1, x2, y2 = cv_clip_line(W, H, x1, y1, x2, y2)
        if not ok:
            return
    dx, dy = x2 - x1, y2 - y1
    sx = sy = 1
    if dx < 0:                                   # leftToRight: start from the end point with the smaller x
        dx, dy = -dx, -dy
        x1, y1 = x2, y2
    if dy < 0:
        dy, sy = -dy, -1
    vert = dy > dx
    if vert:
        dx, dy = dy, dx
    err, plus, minus = dx - 2 * dy, 2 * dx, -2 * dy
    x, y = x1, y1
    for _ in range(dx + 1):
        img[y, x] = True
        m = err < 0
        err += minus + (plus if m else 0)
        if vert:                                 # the major axis is y: always a step in y, a step in x when the error went negative
            y += sy
            x += sx if m else 0
        else:
            x += sx
            y += sy if m else 0


def cv_circle_masks(cx, cy, r, H, W):
    """Circle(img, (cx, cy), r, color, fill) for fill = 1 and fill = 0: (filled, outline) boolean masks.  The midpoint walk over
    one octant; every visited (dx, dy) paints the rows cy +- dy with half-width dx and the rows cy +- dx with half-width dy (fill),
    or the eight symmetric points (outline); everything clipped to the image."""
    fill = np.zeros((H, W), dtype=bool)
    ring = np.zeros((H, W), dtype=bool)

    def hline(y, xa, xb):
        if 0 <= y < H:
            xa, xb = max(xa, 0), min(xb, W - 1)
            if xa <= xb:
                fill[y, xa:xb + 1] = True

    def point(y, x):
        if 0 <= y < H and 0 <= x < W:
            ring[y, x] = True
    err, dx, dy, plus, minus = 0, int(r), 0, 1, (int(r) << 1) - 1
    while dx >= dy:
        for yy, hw in ((cy - dy, dx), (cy + dy, dx), (cy - dx, dy), (cy + dx, dy)):
            hline(yy, cx - hw, cx + hw)
            point(yy, cx - hw)
            point(yy, cx + hw)
        dy += 1
        err += plus
        plus += 2
        if err > 0:                              # mask = (err <= 0) - 1: all ones when err > 0
            err -= minus
            dx -= 1
            minus -= 2
    return fill, ring


def cv_poly_masks(pts, H, W):
    """drawContours(img, [pts], 0, color, -1) and (..., 1) for one closed polygon with integer vertices: (filled, outline)."""
    pts = [(int(x), int(y)) for x, y in np.asarray(pts).reshape(-1, 2)]
    n = len(pts)
    ring = np.zeros((H, W), dtype=bool)
    edges = []                                   # [y0, y1, x (16.16 at y0), dx (16.16 per row)]
    p0 = pts[n - 1]
    for i in range(n):
        p1 = pts[i]
        cv_line(ring, p0, p1)                    # CollectPolyEdges draws the edge (LINE_8) before it stores it
        if p0[1] != p1[1]:
            (xa, ya), (xb, yb) = (p0, p1) if p0[1] < p1[1] else (p1, p0)
            edges.append([ya, yb, xa << XY_SHIFT, _trunc_div((p1[0] - p0[0]) << XY_SHIFT, p1[1] - p0[1])])
        p0 = p1
    fill = ring.copy()
    if not edges:
        return fill, ring
    y_min, y_max = min(e[0] for e in edges), min(max(e[1] for e in edges), H)
    # WHICH OpenCV this restates: the run rule below - ceil of the left edge, floor of the right one, no
    # half-pixel offset on the edges, un-clipped end points - is FillEdgeCollection as OpenCV 2.4 / 3.x / 4.0-4.5.1 wrote it.  Both
    # reviewers recall (from memory; no OpenCV source or wheel exists offline to settle it) that 4.5.2+ passes line_type into
    # FillEdgeCollection, adds XY_ONE >> 1 to edge.x for LINE_8 and rounds both ends (delta = 0), and rebuilds clipped edges from the
    # clipLine() end points.  The reference pins no opencv-python version (requirements.txt), so a current wheel would follow the newer
    # rule.  Because CollectPolyEdges draws every edge FIRST (cv_line above) the two rules can differ only in pixels where a run's end
    # is not on the drawn outline (ties at .5, clipped polygons).  Consequence: g14's masks pin the HIP rasteriser to THIS
    # restatement, not to cv2 - a self-referential pin, as tests/golden/make_golden.py says; Cv2Stub is the one place to swap the day
    # a real cv2 is at hand, and this rule is the first thing to check then.
    for y in range(y_min, y_max):
        # the active edges of FillEdgeCollection at scan line y: y0 <= y < y1, each at x + (y - y0) dx (one addition per scan line
        # since it became active), walked in ascending x; consecutive pairs bound a run, ceil on the left, floor on the right
        act = sorted((e[2] + (y - e[0]) * e[3], e[3]) for e in edges if e[0] <= y < e[1])
        if y < 0:
            continue
        for k in range(0, len(act) - 1, 2):
            xl, xr = act[k][0], act[k + 1][0]
            x1, x2 = (xl + XY_ONE - 1) >> XY_SHIFT, xr >> XY_SHIFT
            if x1 < W and x2 >= 0:
                x1, x2 = max(x1, 0), min(x2, W - 1)
                if x1 <= x2:
                    fill[y, x1:x2 + 1] = True
    return fill, ring


def cv_masks(kind, pts, H, W):
    """(filled, outline) masks of one object under the OpenCV rules above."""
    if kind == CIRCLE:
        return cv_circle_masks(int(pts[0]), int(pts[1]), int(pts[2]), H, W)
    return cv_poly_masks(pts, H, W)


class Cv2Stub:
    """cv2.circle / cv2.boxPoints / cv2.drawContours / cv2.imwrite for driving the reference's generator when golden g14 is made:
    the OpenCV rules restated above (thickness < 0: the filled mask, else the thickness-1 outline)."""

    @staticmethod
    def circle(img, center, radius, color, thickness):
        H, W = img.shape
        fill, ring = cv_circle_masks(int(center[0]), int(center[1]), int(radius), H, W)
        img[fill if thickness < 0 else ring] = color

    @staticmethod
    def boxPoints(rect):
        from be_hip.datagen import box_points
        (cx, cy), (w, h), ang = rect
        return np.array(box_points(cx, cy, w, h, ang), dtype=np.float64)

    @staticmethod
    def drawContours(img, contours, idx, color, thickness):
        H, W = img.shape
        fill, ring = cv_poly_masks(np.asarray(contours[idx], dtype=np.int64), H, W)
        img[fill if thickness < 0 else ring] = color

    @staticmethod
    def imwrite(*a, **k):
        return True


def shape_points(row):
    """(kind, pts) from a `shape` row of be_hip.datagen.draw_scenes."""
    kind = int(row[0])
    if kind == CIRCLE:
        return kind, (int(row[2]), int(row[3]), int(row[4]))
    return kind, np.asarray(row[2:2 + 2 * int(row[1])], dtype=np.int64).reshape(-1, 2)


def blur_kernel(sigma, order=2):
    sigma = max(sigma, 1e-6)                                             # utils/data_generator.py:19-23
    k = int(np.ceil(abs(sigma) * 3))
    x, y = np.meshgrid(np.linspace(-k, k, 2 * k + 1), np.linspace(-k, k, 2 * k + 1))
    psf = np.exp(-np.power((x ** 2 + y ** 2) / (2 * sigma ** 2), order / 2))
    return psf / psf.sum()


def l1_distance(boundary):
    """breadth-first city-block distance to the nearest True pixel; all ones when there is none (:114-125)."""
    if not boundary.any():
        return np.ones(boundary.shape)
    cross = generate_binary_structure(2, 1)
    dist = -np.ones(boundary.shape)
    front = boundary.copy()
    dist[front] = 0
    d = 0
    while (dist == -1).any():
        d += 1
        nxt = binary_dilation(front, structure=cross, iterations=1)
        dist[nxt & ~front] = d
        front = nxt
    return dist


SOBEL_X = np.array([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]], dtype=np.float64)
SOBEL_Y = np.array([[1, 2, 1], [0, 0, 0], [-1, -2, -1]], dtype=np.float64)


def generate_image(shape_rows, prop, nobj, bg, sigmas, H, W, z_far):
    """One image of train_val_data_generator.py:31-130.  shape_rows [MAXO,10] int, prop [MAXO,4] (z,c0,c1,c2) far->near,
    sigmas [MAXO,2].  -> dict(imgs [2,H,W,3] rounded, aif [H,W,3] (0..255), boundary_loc, image_depth, boundary_depth,
    boundary_dist [H,W], deri [2,H,W,3] (/255))."""
    sq = generate_binary_structure(2, 2)
    imgs = np.zeros((2, H, W, 3)) + bg[None, None, None, :]
    aif = np.zeros((H, W, 3)) + bg[None, None, :]
    bloc = np.zeros((H, W))
    idep = np.full((H, W), float(z_far))
    bdep = np.zeros((H, W))
    for o in range(int(nobj)):
        kind, pts = shape_points(shape_rows[o])
        z, col = prop[o, 0], prop[o, 1:]
        m, ol = cv_masks(kind, pts, H, W)
        fill_d = binary_dilation(m, structure=sq, iterations=1)
        ol_d = binary_dilation(ol, structure=sq, iterations=1).astype(np.float64)
        idep[m] = z
        bdep[fill_d] = (ol_d * z)[fill_d]
        mask = m.astype(np.float64) * 255
        for a in range(2):
            mb = convolve(mask, blur_kernel(sigmas[o, a]), mode="reflect")
            w = mb > 0
            for j in range(3):
                imgs[a, :, :, j][w] = mb[w] / 255 * col[j] + (1 - mb[w] / 255) * imgs[a, :, :, j][w]
        bloc[m] = (ol.astype(np.float64) * 255)[m]
        aif[m] = col
    imgs = imgs.round()
    deri = np.zeros_like(imgs)
    for a in range(2):
        deri[a] = np.sqrt(convolve(imgs[a], SOBEL_X[:, :, None]) ** 2 + convolve(imgs[a], SOBEL_Y[:, :, None]) ** 2)
    return dict(imgs=imgs, aif=aif, boundary_loc=bloc, image_depth=idep, boundary_depth=bdep,
                boundary_dist=l1_distance(bloc > 0), deri=deri / 255)


def candidates(boundary_loc, R=21):
    """pixels a patch may be centred on (:214-218)."""
    half = R // 2
    d = binary_dilation(boundary_loc, structure=generate_binary_structure(2, 2), iterations=half + 1)
    margin = np.zeros(boundary_loc.shape, dtype=bool)
    margin[half:-half, half:-half] = True
    return d & margin
