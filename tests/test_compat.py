"""Source-compatibility of the drop-in header set include/sff/: the reference's UNMODIFIED src/main.cpp
(XML parser + main) must compile against it and link to libsffgpu (oracle/Makefile target `compat`),
and the resulting binary must drive the GPU path from an XML config exactly like the Python binding
does for the same seed."""
import os
import subprocess

import numpy as np
import pytest

import common

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_ref", "sff_main")
HAVE_REF = os.path.isdir("/root/reference/src")


@pytest.mark.skipif(not HAVE_REF, reason="reference tree not present on this box")
def test_reference_main_compiles_against_dropin_headers():
    import space_filling_forest_star_amd as S
    if not os.path.exists(S.lib_path()):
        S.build_library()
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "compat"])
    assert os.path.exists(BIN)
    assert subprocess.run([BIN]).returncode == 2                      # src/main.cpp:15-17
    r = subprocess.run([BIN, "/nonexistent.xml"], capture_output=True, text=True)
    assert r.returncode == 1 and "Cannot open config file" in r.stdout  # src/main.cpp:46-49
    # the shipped example config is rejected by the reference's own parser (lazy + priorityBias)
    r = subprocess.run([BIN, "test_2D.xml"], cwd="/root/reference", capture_output=True, text=True)
    assert r.returncode == 1 and "Problem loading error" in r.stdout


def write_obj(path, tri9):
    with open(path, "w") as f:
        k = 1
        for t in np.asarray(tri9).reshape(-1, 9):
            for v in range(3):
                f.write("v %s %s %s\n" % tuple(repr(float(x)) for x in t[3 * v:3 * v + 3]))
            f.write("f %d %d %d\n" % (k, k + 1, k + 2))
            k += 3


def cxx_num(v):
    return "%g" % v   # default ostream formatting (6 significant digits)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(BIN), reason="oracle/_ref/sff_main not built (needs the reference tree)")
@pytest.mark.parametrize("solver,optimize", [("sff", "false"), ("sff", "true"), ("rrt", "false")])
def test_reference_main_drives_the_gpu_path(tmp_path, solver, optimize):
    import space_filling_forest_star_amd as S
    m = common.meshes()
    write_obj(tmp_path / "map.obj", m["triang"])
    write_obj(tmp_path / "robot.obj", m["robot_cylinder_small"])
    pts = common.XML_POINTS["triang"][:4]
    xml = ['<?xml version="1.0" ?>',
           '<Problem solver="%s" optimize="%s" smoothing="false" scale="10">' % (solver, optimize),
           '<Robot file="%s" is_obj="true"/>' % (tmp_path / "robot.obj"),
           '<Environment collision="0.01"><Obstacle file="%s" is_obj="true" position="[0; 0; 0]"/></Environment>'
           % (tmp_path / "map.obj"),
           "<Points>"] + ['<Point coord="[%s; %s; %s]"/>' % tuple(p) for p in (pts if optimize == "false" or solver == "sff" else pts[:1])] + [
           "</Points>",
           '<Range autoDetect="false"><RangeX min="-10" max="10"/><RangeY min="-10" max="10"/><RangeZ min="0" max="10"/></Range>',
           '<Distances dtree="0.5" circum="0.4"/>', '<Thresholds standard="5"/>', '<MaxIterations value="3000"/>',
           '<Save><Tree file="%s" is_obj="false"%s/><Params file="%s" id="compat"/><Goals file="%s" is_obj="false"/>'
           '<RawPath file="%s" is_obj="false"/></Save>'
           % (tmp_path / "tree.tri", ' everyIteration="1000"' if (solver, optimize) == ("sff", "false") else "",
              tmp_path / "params.csv", tmp_path / "goals.tri", tmp_path / "paths.tri"),
           "</Problem>"]
    (tmp_path / "cfg.xml").write_text("\n".join(xml))
    env = dict(os.environ, SFF_SEED="21", SFF_WAVE="64")
    r = subprocess.run([BIN, str(tmp_path / "cfg.xml")], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Saving trees" in r.stdout
    # the same run through the Python binding
    sc = common.scenario("triang")
    ctx = S.Context(0)
    ctx.upload_env(sc["env"])
    ctx.upload_robot(sc["robot"])
    roots = sc["xml_points"][:4]
    if solver == "sff":
        f = S.Forest(ctx, roots, sc["limits"], dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"], dim=6,
                     optimize=(optimize == "true"), max_iterations=3000, wave=64, seed=21)
        f.run()
        n = f.nodes()
        tree_of = n["tree"]
        iters = f.stats()["iterations"]
    else:
        f = S.Rrt(ctx, roots, sc["limits"], dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"], dim=6,
                  max_iterations=3000, seed=21)
        f.run()
        n = f.nodes()
        tree_of = n["root_tree"]
        iters = f.stats()["iterations"]
    # tree dump: per tree (creation order), per node (insertion order): child pose, parent pose, tree id, iteration
    want = ["#X1 Y1 Z1 Yaw1 Pitch1 Roll1 X2 Y2 Z2 Yaw2 Pitch2 Roll2 TreeID IterationOfCreation"]
    for t in range(int(tree_of.max()) + 1):
        for i in np.where(tree_of == t)[0]:
            if n["cost"][i] != 0:
                a = n["pos"][i].copy(); b = n["pos"][n["parent"][i]].copy()
                a[:3] /= 10.0; b[:3] /= 10.0
                want.append(" ".join([cxx_num(v) for v in a] + [cxx_num(v) for v in b] + [str(t), str(int(n["iter"][i]))]))
    got = (tmp_path / "tree.tri").read_text().strip().split("\n")
    assert len(got) == len(want)
    assert got == want
    row = (tmp_path / "params.csv").read_text().strip().split(",")
    assert row[0] == "compat" and int(row[2]) == iters
    if (solver, optimize) == ("sff", "false"):
        # saveIterCheck (src/problemStruct.h:256-261): "iter_<k>_" dumps = the forest at the end of the wave in which
        # iteration k fell: every node created up to iteration k is in it, and it is a subset of the final dump
        final = set(got[1:])
        for k in (1000, 2000):
            dump = (tmp_path / ("iter_%d_tree.tri" % k)).read_text().strip().split("\n")
            assert dump[0] == want[0]
            lines = set(dump[1:])
            assert lines <= final and len(lines) < len(final)
            assert {l for l in final if int(l.split()[-1]) <= k} <= lines
    f.close()
    ctx.close()
