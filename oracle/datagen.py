"""Oracle: the synthetic-shape data generator on the CPU (TEST INFRASTRUCTURE).

Two parts:
 * the rasterisation RULE of the build (be_hip/datagen.py docstring) in numpy -- `inside_mask` / `outline_mask` -- and a
   stand-in for the three cv2 drawing calls the reference makes (cv2 is not installed and its rasteriser is not
   specified bit for bit), used ONLY to drive the reference's own generator when golden g14 is made;
 * a restatement of train_val_data_generator.py:31-130 (scene -> images, all-in-focus, boundary / depth maps,
   city-block distances, Sobel maps), :165-182 (noise model, deterministic part) and :226-252 (patch crop + in-patch
   distance transform) with the same scipy.ndimage calls, pinned against the reference by golden g14.
"""
import numpy as np
from scipy.ndimage import binary_dilation, convolve, generate_binary_structure

CIRCLE, RECT, TRI = 0, 1, 2


def _inside(kind, pts, X, Y):
    """analytic inside test at integer coordinate arrays X, Y (may lie outside the image)."""
    if kind == CIRCLE:
        cx, cy, r = pts
        return (X - cx) ** 2 + (Y - cy) ** 2 <= r * r
    pts = np.asarray(pts, dtype=np.int64).reshape(-1, 2)
    nv = pts.shape[0]
    pos = np.ones(X.shape, dtype=bool)
    neg = np.ones(X.shape, dtype=bool)
    for k in range(nv):
        x0, y0 = pts[k]
        x1, y1 = pts[(k + 1) % nv]
        c = (x1 - x0) * (Y - y0) - (y1 - y0) * (X - x0)
        pos &= c >= 0
        neg &= c <= 0
    box = (X >= pts[:, 0].min()) & (X <= pts[:, 0].max()) & (Y >= pts[:, 1].min()) & (Y <= pts[:, 1].max())
    return (pos | neg) & box


def inside_mask(kind, pts, H, W):
    Y, X = np.meshgrid(np.arange(H, dtype=np.int64), np.arange(W, dtype=np.int64), indexing="ij")
    return _inside(kind, pts, X, Y)


def outline_mask(kind, pts, H, W):
    Y, X = np.meshgrid(np.arange(H, dtype=np.int64), np.arange(W, dtype=np.int64), indexing="ij")
    m = _inside(kind, pts, X, Y)
    full = _inside(kind, pts, X - 1, Y) & _inside(kind, pts, X + 1, Y) & _inside(kind, pts, X, Y - 1) & _inside(kind, pts, X, Y + 1)
    return m & ~full


class Cv2Stub:
    """cv2.circle / cv2.boxPoints / cv2.drawContours / cv2.imwrite with the build's rasterisation rule."""

    @staticmethod
    def circle(img, center, radius, color, thickness):
        H, W = img.shape
        m = (inside_mask if thickness < 0 else outline_mask)(CIRCLE, (int(center[0]), int(center[1]), int(radius)), H, W)
        img[m] = color

    @staticmethod
    def boxPoints(rect):
        from be_hip.datagen import box_points
        (cx, cy), (w, h), ang = rect
        return np.array(box_points(cx, cy, w, h, ang), dtype=np.float64)

    @staticmethod
    def drawContours(img, contours, idx, color, thickness):
        H, W = img.shape
        pts = np.asarray(contours[idx], dtype=np.int64)
        kind = RECT if pts.shape[0] == 4 else TRI
        img[(inside_mask if thickness < 0 else outline_mask)(kind, pts, H, W)] = color

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
        m = inside_mask(kind, pts, H, W)
        ol = outline_mask(kind, pts, H, W)
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
