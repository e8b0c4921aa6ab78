"""Visualizer: the 2 x 5 result sheet the evaluation scripts save per image pair (utils/visualization.py:5-63 of the
reference; called at blurry_edges_test.py:157-168).  Drawing only - not on the compute path.  cv2 does the colour
maps, nearest-neighbour scaling and captions, exactly as in the reference, and is imported when the class is
constructed (the evaluation scripts import cv2 themselves), so `import utils` works on machines without it."""
import numpy as np

# (row, column) of each panel on the sheet and its caption; visualize() takes the panels in this order
_PANELS = [("Noisy input 1", 0, 0), ("Noisy input 2", 1, 0), ("Restored colormap 1", 0, 1), ("Restored colormap 2", 1, 1),
           ("Sharpened colormap", 0, 2), ("Refocused colormap *", 1, 2), ("Confidence map", 0, 3),
           ("Estimated boundary map", 1, 3), ("Ground truth depth map", 0, 4), ("Estimated depth map", 1, 4)]
_Z_LO, _Z_SPAN = 0.73, 0.45            # depth range of the rainbow bar in metres


class Visualizer:
    def __init__(self, rho_prime, img_size=147, gap_v=20, gap_h=5, scale=10, fontsize_scale=0.35):
        import cv2
        self.cv2 = cv2
        self.rho_prime, self.img_size = rho_prime, img_size
        self.gap_v, self.gap_h, self.scale, self.fontsize_scale = gap_v, gap_h, scale, fontsize_scale
        self.colormap_f = self.get_color_map()
        self.canvas_blank = self.get_blank_canvas()

    def _origin(self, row, col):
        """top-left pixel of a panel"""
        s = self.scale
        return (self.gap_v + row * (self.img_size + self.gap_v)) * s, col * (self.img_size + self.gap_h) * s

    def _text(self, canvas, txt, x, y, size=1.0):
        self.cv2.putText(canvas, txt, (x, y), self.cv2.FONT_HERSHEY_SIMPLEX, self.fontsize_scale * self.scale * size,
                         (0, 0, 0), self.scale)

    def get_color_map(self):
        lut = np.zeros((256, 1, 3), dtype=np.uint8)
        lut[:, 0, 1] = np.arange(256)                       # confidence: black -> green
        return lut

    def get_blank_canvas(self):
        cv2, s, n, gv, gh = self.cv2, self.scale, self.img_size, self.gap_v, self.gap_h
        canvas = np.full(((2 * n + 3 * gv) * s, (5 * n + 5 * gh + 40) * s, 3), 255.0)
        ramp = (np.linspace(1, 0, 1000)[:, None] * 0.43 + 0.02) / 0.45
        bar = cv2.applyColorMap((ramp * 255).clip(0, 255).astype(np.uint8), cv2.COLORMAP_RAINBOW)
        x0 = (5 * n + 5 * gh) * s
        canvas[gv * s:(2 * n + 2 * gv) * s, x0:x0 + 2 * gh * s] = cv2.resize(bar, (2 * gh * s, (2 * n + gv) * s),
                                                                            interpolation=cv2.INTER_NEAREST)
        self._text(canvas, '75', (5 * n + 8 * gh) * s, (2 * n + int(gv * 1.9)) * s)
        self._text(canvas, '118', (5 * n + int(gh * 7.6)) * s, int(gv * 1.6) * s)
        self._text(canvas, 'cm', (5 * n + int(gh * 7.6)) * s, int(gv * 0.7) * s)
        for caption, row, col in _PANELS:
            y, x = self._origin(row, col)
            self._text(canvas, caption, x, y - gv * s + int(gv * 0.7) * s)
        self._text(canvas, f'* Refocused with a reference of optical power: {self.rho_prime}', (2 * n + 2 * gh) * s,
                   (2 * n + int(gv * 2.7)) * s, size=0.8)
        return canvas

    def visualize(self, I_1, I_2, C_1, C_2, C_shpd, C_refoc, F, B, Z_gt, Z):
        cv2, s, n = self.cv2, self.scale, self.img_size
        rainbow = lambda z: cv2.applyColorMap(((z - _Z_LO) / _Z_SPAN * 255).clip(0, 255).astype(np.uint8), cv2.COLORMAP_RAINBOW)
        conf = cv2.applyColorMap((F * 255).clip(0, 255).astype(np.uint8), self.colormap_f)
        est = rainbow(Z)
        est[(est[:, :, 0] == 0) & (est[:, :, 1] == 0) & (est[:, :, 2] == 255)] = 0      # no estimate -> black instead of the bar's end colour
        panels = [I_1 * 255, I_2 * 255, C_1 * 255, C_2 * 255, C_shpd * 255, C_refoc * 255, conf, (B * 255).clip(0, 255),
                  rainbow(Z_gt), est]
        canvas = self.canvas_blank.copy().astype(np.uint8)
        for img, (_, row, col) in zip(panels, _PANELS):
            y, x = self._origin(row, col)
            big = cv2.resize(img, (n * s, n * s), interpolation=cv2.INTER_NEAREST)
            canvas[y:y + n * s, x:x + n * s, :] = big if big.ndim == 3 else big[:, :, None]
        return canvas
