"""DataGenerator: thin-lens blur radius and PSF of the two-aperture camera (utils/data_generator.py:3-23 of the reference).
Host-side constants only; the image synthesis itself runs on the GPU (be_hip/datagen.py)."""
import numpy as np

_CAMERA_FIELDS = (("s", "s"), ("Sigma_cam", "sigma_cam"), ("pixel_pitch", "pixel_pitch"))


class DataGenerator:
    def __init__(self, args):
        camera = args.cam_params
        for attr, key in _CAMERA_FIELDS:
            setattr(self, attr, camera[key])
        self.rhos = np.array([camera['rho_1'], camera['rho_2']])       # optical power of the two apertures
        self.n_img = self.rhos.size
        self.mag = args.mag
        self.data_path, self.Z_range, self.alpha, self.sigma = args.data_path, args.Z_range, args.alpha, args.sigma

    def get_kernel_sigma(self, z):
        """Blur radius in pixels, one per aperture, of an object z metres away: |(1/z - rho) s + 1| Sigma / pitch / mag."""
        defocus = np.abs((1.0 / z - self.rhos) * self.s + 1.0)
        return defocus * self.Sigma_cam / self.pixel_pitch / self.mag

    def get_blur_kernel(self, sigma, order=2):
        """(2k+1)^2 generalised-Gaussian PSF, k = ceil(3 sigma), normalised to unit sum (order 2 = Gaussian)."""
        sigma = max(sigma, 1e-6)
        half = int(np.ceil(3 * abs(sigma)))
        taps = np.arange(-half, half + 1, dtype=np.float64)
        r2 = taps[:, None] ** 2 + taps[None, :] ** 2
        weights = np.exp(-((r2 / (2 * sigma ** 2)) ** (order / 2)))
        return weights / weights.sum()
