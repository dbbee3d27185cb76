#!/usr/bin/env python3
"""Parse the reference's DATA files (maps/, models/; BSD-3, see tests/golden/DATA_LICENSE) with the
oracle's restatement of the reference parser (src/environment.h:125-223 quirks kept) and store
the triangle arrays (scale 1, position 0) as compressed .npz fixtures.  Runs only where
/root/reference exists; the fixtures travel to the GPU box instead of the data files."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle_lib as O
REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
if not os.path.isdir(REF):
    sys.exit("reference tree not present")
items = {
    "dense_3D": ("maps/dense_3D.obj", True),
    "triang": ("models/3D/triang.obj", True),
    "building": ("maps/building.obj", True),
    "robot_cylinder_small": ("models/3D/robot_cylinder_small.obj", True),
    "dense_2D": ("maps/dense.tri", False),
    "robot_small_2D": ("models/robot_small.obj", True),
}
arrs = {}
for name, (rel, is_obj) in items.items():
    p = os.path.join(REF, rel)
    t = O.parse_obj(p) if is_obj else O.parse_tri2d(p)
    arrs[name] = t
    print(name, t.shape, t.min(axis=0)[:3], t.max(axis=0)[:3])
np.savez_compressed(os.path.join(OUT, "meshes.npz"), **arrs)
print(os.path.getsize(os.path.join(OUT, "meshes.npz")), "bytes")
