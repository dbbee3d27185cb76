"""Shared helpers for the tests: mesh fixtures, seeded roots, scenario definitions."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_MESH = None


def meshes():
    global _MESH
    if _MESH is None:
        _MESH = dict(np.load(os.path.join(ROOT, "tests", "golden", "meshes.npz")))
    return _MESH


def scaled(tri9, scale):
    return np.ascontiguousarray(tri9 * scale)


# name -> (env mesh, robot mesh, scale, limits, dist_tree, sampling_dist, dim)   (distances already scaled)
SCENARIOS = {
    # C3/C4 of BASELINE.json: dense_3D.obj, 6-DoF, step chosen so that a 1M-node budget is reachable
    # (with the 2-D file's circum=80 the forest saturates at ~5.5k nodes)
    "dense3d": ("dense_3D", "robot_cylinder_small", 1.0, [-60, 2060, -60, 2110, 0, 1000], 18.0, 14.0, 6),
    # the 2-D file's own distances on the 3-D map: saturating, many borders
    "dense3d_coarse": ("dense_3D", "robot_cylinder_small", 1.0, [-60, 2060, -60, 2110, 0, 1000], 100.0, 80.0, 6),
    # C2: test_triang.xml geometry (scale 10, dtree 0.5, circum 0.4, ranges +-10 / 0..10)
    "triang": ("triang", "robot_cylinder_small", 10.0, [-100, 100, -100, 100, 0, 100], 5.0, 4.0, 6),
    # C5: test_building.xml geometry (building.obj 26 908 tris, scale 10, ranges +-7 / 0..14)
    "building": ("building", "robot_cylinder_small", 10.0, [-70, 70, -70, 70, 0, 140], 5.0, 4.0, 6),
    # C1: test_2D.xml geometry (dense.tri, 2-D robot)
    "dense2d": ("dense_2D", "robot_small_2D", 1.0, [-60, 2060, -60, 2110, 0, 0], 100.0, 80.0, 2),
}


# start points the reference's example XMLs list (test_triang.xml, test_2D.xml), un-scaled
XML_POINTS = {
    "triang": [[-1.5, 4, 3], [2.9, 0.3, 7], [2.7, -3.4, 5], [-3.96, -2.4, 1], [4.2, 3.5, 1], [-4.3, 3.5, 8]],
    "building": [[-5.376207930019596, -5.338214644384135, 0.7519141368058615],
                 [5.309629695920314, -5.3435510614853206, 2.225339932086428],
                 [-5.445894591081855, 4.942949264187298, 8.034669391216193],
                 [3.4004901456614443, 0.3800196290706541, 10], [0.30319967716688234, 0.057430173261578954, 7]],
    "dense2d": [[1500, 1600, 0], [100, 100, 0], [500, 1700, 0], [1440, 330, 0]],
}


def scenario(name):
    env, rob, scale, lim, dt, sd, dim = SCENARIOS[name]
    m = meshes()
    sc = dict(env=scaled(m[env], scale), robot=scaled(m[rob], scale), limits=lim, dist_tree=dt, sampling_dist=sd,
              dim=dim, scale=scale, xml_points=None)
    if name in XML_POINTS:
        pts = np.zeros((len(XML_POINTS[name]), 6))
        pts[:, :3] = np.array(XML_POINTS[name], dtype=np.float64) * scale
        sc["xml_points"] = pts
    return sc


def free_roots(collide_fn, limits, n, seed=1, dim=6):
    """n seeded uniform points in the limits that do not collide (orientation zero, like the XML points)."""
    rs = np.random.RandomState(seed)
    out = []
    while len(out) < n:
        p = np.array([rs.uniform(limits[0], limits[1]), rs.uniform(limits[2], limits[3]),
                      rs.uniform(limits[4], limits[5]) if dim == 6 else 0.0, 0, 0, 0])
        if not collide_fn(p):
            out.append(p)
    return np.array(out)


def random_poses(limits, n, seed, dim=6):
    rs = np.random.RandomState(seed)
    p = np.zeros((n, 6))
    p[:, 0] = rs.uniform(limits[0], limits[1], n)
    p[:, 1] = rs.uniform(limits[2], limits[3], n)
    if dim == 6:
        p[:, 2] = rs.uniform(limits[4], limits[5], n)
        p[:, 3:] = rs.uniform(-np.pi, np.pi, (n, 3))
    return p


def poses_near_surface(env_tri9, n, seed, spread, dim=6):
    """poses scattered around random points ON the environment triangles (guarantees many contacts)"""
    rs = np.random.RandomState(seed)
    t = env_tri9[rs.randint(0, len(env_tri9), n)].reshape(n, 3, 3)
    w = rs.dirichlet([1, 1, 1], n)
    pts = (t * w[:, :, None]).sum(axis=1)
    p = np.zeros((n, 6))
    p[:, :3] = pts + rs.normal(0, spread, (n, 3))
    if dim == 6:
        p[:, 3:] = rs.uniform(-np.pi, np.pi, (n, 3))
    else:
        p[:, 2] = 0
    return p
