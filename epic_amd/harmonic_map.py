"""Occupancy-grid loaders: image file -> ``Harmonic`` arrays.

``HarmonicMap.load`` follows the reference's python rule (libepic/python/epic/harmonic_map.py:54-100):
m = [H, W]; pixel 255 -> goal (u = 0, locked); pixel 0 -> obstacle (u = -1e6, locked); anything else -> free
(u = -1e6, unlocked).  PIL replaces cv2 (absent here); the OpenCV click-to-streamline viewer is not part of the
relaxation path and is not reproduced.
"""
import numpy as np

from .harmonic import Harmonic

LOG_SPACE_GOAL = np.float32(0.0)
LOG_SPACE_OBSTACLE = np.float32(-1e6)
LOG_SPACE_FREE = np.float32(-1e6)


def grid_from_gray(px):
    """uint8 gray image -> (m, u, locked) by the reference rule."""
    px = np.asarray(px)
    if px.ndim != 2:
        raise ValueError("expected a 2-d grayscale image")
    u = np.where(px == 255, LOG_SPACE_GOAL, LOG_SPACE_FREE).astype(np.float32)
    locked = ((px == 0) | (px == 255)).astype(np.uint32)
    return [int(px.shape[0]), int(px.shape[1])], u, locked


class HarmonicMap(Harmonic):
    """A 2-d ``Harmonic`` loaded from a grayscale image."""

    def __init__(self):
        super().__init__()
        self.image = None

    def load(self, filename):
        from PIL import Image

        self.image = np.array(Image.open(filename).convert("L"))
        m, u, locked = grid_from_gray(self.image)
        self.set_grid(m, u, locked)
        return self
