"""DataGenerator: thin-lens blur radius and PSF of the two-aperture camera (utils/data_generator.py:3-23 of the reference).
Host-side constants only; the image synthesis itself runs on the GPU (be_hip/datagen.py)."""
import numpy as np


class DataGenerator:
    def __init__(self, args):
        cam = args.cam_params
        self.data_path, self.Z_range = args.data_path, args.Z_range
        self.s, self.rhos = cam['s'], np.array([cam['rho_1'], cam['rho_2']])
        self.Sigma_cam, self.pixel_pitch, self.mag = cam['sigma_cam'], cam['pixel_pitch'], args.mag
        self.alpha, self.sigma = args.alpha, args.sigma
        self.n_img = len(self.rhos)

    def get_kernel_sigma(self, z):
        """Blur radius in pixels of each aperture for an object at distance z (m): |(1/z - rho) s + 1| Sigma / pitch / mag."""
        return np.abs((1 / z - self.rhos) * self.s + 1) * self.Sigma_cam / self.pixel_pitch / self.mag

    def get_blur_kernel(self, sigma, order=2):
        """(2k+1)^2 generalised-Gaussian PSF, k = ceil(3 sigma), normalised to unit sum (order 2 = Gaussian)."""
        sigma = max(sigma, 1e-6)
        half = int(np.ceil(3 * abs(sigma)))
        taps = np.arange(-half, half + 1, dtype=np.float64)
        r2 = taps[:, None] ** 2 + taps[None, :] ** 2
        weights = np.exp(-((r2 / (2 * sigma ** 2)) ** (order / 2)))
        return weights / weights.sum()
