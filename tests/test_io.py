"""JSON export / YAML config import, mirroring the reference's test/json.jl and test/yaml.jl
(fixtures tests/golden/yaml/t1.yml, t2.yml are the reference's own test data files)."""
import io
import json
import math
import os

import numpy as np

import ransac_jl_amd as R

HERE = os.path.dirname(os.path.abspath(__file__))


def test_todict_and_exportjson():  # test/json.jl:3-80
    p1 = np.array([15.6, 0, -13.7])
    n1 = np.array([34, 45, 7.0]) / np.linalg.norm([34, 45, 7.0])
    s_plane = R.FittedPlane(p1, n1)
    d_plane = {"type": "plane", "point": list(p1), "normal": list(n1)}
    assert R.toDict(s_plane) == d_plane
    s_sphere = R.FittedSphere(p1, 13.23444, True)
    d_sphere = {"type": "sphere", "radius": 13.23444, "center": list(p1), "outwards": True}
    assert R.toDict(s_sphere) == d_sphere
    a1 = np.array([-17.1, 8, 2.42])
    s_cyl = R.FittedCylinder(a1, p1, 0.13, False)
    d_cyl = {"type": "cylinder", "axis": list(a1), "center": list(p1), "radius": 0.13, "outwards": False}
    assert R.toDict(s_cyl) == d_cyl
    ax1 = np.array([-1.5, 7, 2]) / np.linalg.norm([-1.5, 7, 2])
    s_cone = R.FittedCone([0.0, 0, 0], ax1, 0.785, True)
    d_cone = {"type": "cone", "apex": [0.0, 0.0, 0.0], "axis": list(ax1), "opang": 0.785, "outwards": True}
    assert R.toDict(s_cone) == d_cone
    sc1, ss1 = R.ExtractedShape(s_plane, [1]), R.ExtractedShape(s_cone, [1, 2, 3])
    assert R.toDict(sc1) == d_plane and R.toDict(ss1) == d_cone
    assert R.toDict([s_plane, s_sphere, s_cyl, s_cone]) == {"primitives": [d_plane, d_sphere, d_cyl, d_cone]}
    assert R.toDict([sc1, ss1]) == {"primitives": [d_plane, d_cone]}
    buf = io.StringIO()
    R.exportJSON(buf, s_cone, 2)
    assert json.loads(buf.getvalue()) == d_cone
    buf = io.StringIO()
    R.exportJSON(buf, [s_plane, s_cone])
    assert json.loads(buf.getvalue()) == {"primitives": [d_plane, d_cone]}


def test_readconfig_t1():  # test/yaml.jl:58-70
    conf = R.readconfig(os.path.join(HERE, "golden", "yaml", "t1.yml"))
    p1 = R.ransacparameters()
    p1 = R.ransacparameters(p1, sphere={"ϵ": 0.2, "α": 0.05, "sphere_par": 0.01}, plane={"ϵ": 0.1, "α": 0.01})
    p1 = R.ransacparameters(p1, cylinder={"α": 0.0872}, cone={"ϵ": 1, "α": 3.14, "minconeopang": 1.0})
    p1 = R.ransacparameters(p1, iteration={"drawN": 9, "minsubsetN": 2, "prob_det": 0.999, "τ": 10000,
                                           "itermax": 100000, "shape_types": [R.FittedPlane, R.FittedSphere]})
    p1 = R.ransacparameters(p1, common={"parallelthrdeg": 0.5, "collin_threshold": 0.3})
    assert conf == p1
    c = R.params_to_c(conf)
    assert (c.drawN, c.minsubsetN, c.tau, c.itermax, c.n_shape_types) == (9, 2, 10000, 100000, 2)
    assert c.eps[R.SPHERE] == 0.2 and c.cos_alpha[R.PLANE] == math.cos(0.01) and c.minconeopang == 1.0


def test_readconfig_t2():  # test/yaml.jl:72-83
    conf = R.readconfig(os.path.join(HERE, "golden", "yaml", "t2.yml"))
    p1 = R.ransacparameters()
    p1 = R.ransacparameters(p1, plane={"ϵ": 0.35, "α": 1.0872}, sphere={"sphere_par": 0.025})
    p1 = R.ransacparameters(p1, iteration={"itermax": 100})
    p1 = R.ransacparameters(p1, common={"collin_threshold": 0.22, "parallelthrdeg": 1.2})
    assert conf == p1
