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

    def compute_streamline(self, x, y, step_size=0.2, cd_precision=0.4, max_length=int(1e6)):
        """The way-points from the "float pixel" (x, y) to a goal on the host copy of the relaxed field, as a list of
        (x, y) tuples: the reference's HarmonicMap._compute_streamline (harmonic_map.py:103-131), same defaults, through
        harmonic_compute_path_2d_cpu / harmonic_free_path_cpu.  Raises RuntimeError with the library's code on failure."""
        import ctypes as ct

        from . import epic_harmonic as eh

        k = ct.c_uint(0)
        raw = ct.POINTER(ct.c_float)()
        rc = eh._epic.harmonic_compute_path_2d_cpu(self, float(x), float(y), float(step_size), float(cd_precision),
                                                   int(max_length), ct.byref(k), ct.byref(raw))
        if rc != 0:
            raise RuntimeError("harmonic_compute_path_2d_cpu failed with code %d" % rc)
        path = [(raw[2 * i], raw[2 * i + 1]) for i in range(int(k.value))]
        eh._epic.harmonic_free_path_cpu(ct.byref(raw))
        return path

    _compute_streamline = compute_streamline  # the reference's (private) name


# ---- the ROS ingestion routes ------------------------------------------------------------------
# maps/*.yaml + image --map_server--> nav_msgs/OccupancyGrid --navigation node--> Harmonic arrays, and
# costmap_2d --nav_core plugin--> Harmonic arrays.  map_server is third-party (ros-planning/navigation, not under the
# reference tree); its published trinary rule is restated here.  The two node-side rules are the reference's.

OCCUPANCY_OBSTACLE_THRESHOLD = 50    # include/epic/epic_navigation_node_constants.h: EPIC_OCCUPANCY_GRID_OBSTACLE_THRESHOLD
OCCUPANCY_NO_CHANGE = -2             # EPIC_OCCUPANCY_GRID_NO_CHANGE
COSTMAP_OBSTACLE_THRESHOLD = 250     # src/epic_nav_core_plugin.cpp:48


def load_map_yaml(path):
    """maps/*.yaml (image, resolution, origin, occupied_thresh, free_thresh, negate)."""
    import os

    import yaml

    with open(path) as f:
        meta = yaml.safe_load(f)
    meta["image_path"] = os.path.join(os.path.dirname(os.path.abspath(path)), meta["image"])
    return meta


def occupancy_from_image(px, occupied_thresh=0.65, free_thresh=0.196, negate=0):
    """map_server's trinary interpretation of a gray image -> int8 occupancy data in OccupancyGrid order:
    p = (255 - v)/255 (v/255 when negate); p > occupied_thresh -> 100, p < free_thresh -> 0, otherwise -1 (unknown);
    grid row 0 is the BOTTOM row of the image."""
    px = np.asarray(px, dtype=np.float64)
    if px.ndim == 3:                         # colour image: map_server averages the colour channels
        px = px[..., :3].mean(axis=2)
    p = px / 255.0 if negate else (255.0 - px) / 255.0
    occ = np.full(px.shape, -1, dtype=np.int8)
    occ[p > occupied_thresh] = 100
    occ[p < free_thresh] = 0
    return occ[::-1].copy()


def grid_from_occupancy(occ, goals=(), previous=None):
    """The navigation node's /map callback (src/epic_navigation_node_harmonic.cpp:383-422, border rule :294-306):
    interior cells >= 50 become obstacles, -2 leaves the cell as it was, everything else (free 0 and unknown -1) becomes
    free; goal cells are not touched by the map; the border is always an obstacle.  `previous` = (u, locked) of the
    state before this message (the node starts from u = 0, locked = 0: :219-229).  Returns (m, u, locked)."""
    occ = np.asarray(occ)
    rows, cols = occ.shape
    if previous is None:
        u = np.zeros((rows, cols), dtype=np.float32)
        locked = np.zeros((rows, cols), dtype=np.uint32)
    else:
        u = np.array(previous[0], dtype=np.float32).reshape(rows, cols)
        locked = np.array(previous[1], dtype=np.uint32).reshape(rows, cols)
    is_goal = (locked == 1) & (u == LOG_SPACE_GOAL)
    for x, y in goals:
        is_goal[y, x] = True
    inner = np.zeros((rows, cols), dtype=bool)
    inner[1:-1, 1:-1] = True
    change = inner & (occ != OCCUPANCY_NO_CHANGE) & ~is_goal
    obstacle = change & (occ >= OCCUPANCY_OBSTACLE_THRESHOLD)
    free = change & ~obstacle
    u[obstacle] = LOG_SPACE_OBSTACLE
    locked[obstacle] = 1
    u[free] = LOG_SPACE_FREE
    locked[free] = 0
    for x, y in goals:
        u[y, x] = LOG_SPACE_GOAL
        locked[y, x] = 1
    for edge in (np.s_[0, :], np.s_[-1, :], np.s_[:, 0], np.s_[:, -1]):
        u[edge] = LOG_SPACE_OBSTACLE
        locked[edge] = 1
    return [rows, cols], u, locked


def grid_from_costmap(cost, goals=()):
    """The nav_core plugin's costmap rule (src/epic_nav_core_plugin.cpp:153-163, border :167-187): cost >= 250 is an
    obstacle, everything else free, border obstacle; goals (set by makePlan, :341-366) are locked at u = 0."""
    cost = np.asarray(cost)
    rows, cols = cost.shape
    obstacle = cost >= COSTMAP_OBSTACLE_THRESHOLD
    u = np.full((rows, cols), LOG_SPACE_FREE, dtype=np.float32)
    locked = obstacle.astype(np.uint32)
    u[obstacle] = LOG_SPACE_OBSTACLE
    for edge in (np.s_[0, :], np.s_[-1, :], np.s_[:, 0], np.s_[:, -1]):
        u[edge] = LOG_SPACE_OBSTACLE
        locked[edge] = 1
    for x, y in goals:
        u[y, x] = LOG_SPACE_GOAL
        locked[y, x] = 1
    return [rows, cols], u, locked


def load_yaml_map(path, goals=()):
    """maps/*.yaml -> HarmonicMap through the map_server + navigation-node rules."""
    from PIL import Image

    meta = load_map_yaml(path)
    px = np.array(Image.open(meta["image_path"]).convert("L"))
    occ = occupancy_from_image(px, meta.get("occupied_thresh", 0.65), meta.get("free_thresh", 0.196), meta.get("negate", 0))
    m, u, locked = grid_from_occupancy(occ, goals)
    h = HarmonicMap()
    h.image = px
    h.meta = meta
    h.set_grid(m, u, locked)
    return h
