"""C-ABI library: loads, exports every declared symbol, host-side logic (no compute calls without a GPU)."""
import ctypes
import json
import os

import numpy as np
import pytest

from envs import build_oracle_env, build_product_env, spec_for

ROBOTS = ["panda", "ur5", "fetch", "baxter"]


def test_library_exports_every_declared_symbol(vamp):
    from vamp_mvt_amd import _lib

    names = _lib.declared_symbols()
    assert len(names) >= 35
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), n
    assert vamp.abi_version() == 1


def test_no_torch_types_in_the_abi():
    here = os.path.dirname(os.path.abspath(__file__))
    text = open(os.path.join(here, "..", "include", "vamp_mvt_amd.h")).read()
    assert "torch" not in text and "at::" not in text and 'extern "C"' in text


def test_robot_constants_match_oracle_and_models(vamp, oracle):
    here = os.path.dirname(os.path.abspath(__file__))
    assert vamp.robots() == ROBOTS and list(vamp.robots) == ROBOTS
    for name in ROBOTS:
        mod = getattr(vamp, name)
        rid = oracle.robot(name)
        model = json.load(open(os.path.join(here, "..", "vamp_mvt_amd", "robots", f"{name}.json")))
        assert mod.dimension() == oracle.dimension(rid) == model["dimension"]
        assert mod.n_spheres() == oracle.n_spheres(rid) == model["n_spheres"]
        assert mod.resolution() == oracle.resolution(rid) == model["resolution"]
        assert mod.joint_names() == model["joint_names"]
        assert mod.end_effector() == model["end_effector"]
        lo, span = oracle.bounds(rid)
        assert np.array_equal(mod.lower_bounds(), lo)
        assert np.array_equal(mod.upper_bounds(), (lo + span).astype(np.float32))
        mn, mx = mod.min_max_radii()
        assert np.float32(mn) == np.float32(model["min_radius"]) and np.float32(mx) == np.float32(model["max_radius"])


def test_robot_models_are_consistent():
    here = os.path.dirname(os.path.abspath(__file__))
    expect = {"panda": (59, 11, 21, 690), "ur5": (40, 17, 55, 383), "fetch": (111, 15, 48, 2586),
              "baxter": (75, 33, 349, 1845)}
    for name, (n_fine, n_env, n_self, n_pairs) in expect.items():
        m = json.load(open(os.path.join(here, "..", "vamp_mvt_amd", "robots", f"{name}.json")))
        assert m["n_spheres"] == n_fine and len(m["env_groups"]) == n_env
        assert len(m["self_groups"]) == n_self and sum(len(g["pairs"]) for g in m["self_groups"]) == n_pairs
        fine = sorted(s for g in m["env_groups"] for s in g["fine"])
        assert fine == list(range(n_fine))  # every fine sphere is in exactly one link group
        assert sorted(g["bound"] for g in m["env_groups"]) == list(range(n_fine, n_fine + n_env))


@pytest.mark.parametrize("kind", ["shell64", "mixed", "capt"])
def test_host_tables_match_oracle(vamp, oracle, kind):
    """min_distance values, z-aligned classification and CAPT arrays: product host code vs oracle, bit for bit."""
    spec = spec_for(kind, "panda")
    pe, oe = build_product_env(spec), build_oracle_env(oracle, spec)
    t = pe.host_tables()

    def srt(a):  # the getters return the sorted lists (stable by min_distance), like the oracle's
        assert len(a) == 0 or (np.diff(a[:, -1]) >= 0).all()
        return a

    assert np.array_equal(srt(t["spheres"]), oe.spheres())
    assert np.array_equal(srt(t["cuboids"]), oe.cuboids(False))
    assert np.array_equal(srt(t["z_cuboids"]), oe.cuboids(True))
    assert np.array_equal(srt(t["capsules"]), oe.capsules(False))
    assert np.array_equal(srt(t["z_capsules"]), oe.capsules(True))
    assert len(t["capt"]) == oe.counts()[5]
    for i, a in enumerate(t["capt"]):
        b = oe.capt(i)
        assert a["nlog2"] == b["nlog2"]
        assert np.array_equal(a["tests"], b["tests"], equal_nan=True)
        assert np.array_equal(a["aff_starts"], b["aff_starts"])
        assert np.array_equal(a["aabbs"], b["aabbs"])
        assert np.array_equal(a["aff"], b["aff"])
        assert np.array_equal(a["aabb_top"], b["aabb_top"])


def test_shape_constructors(vamp):
    s = vamp.Sphere([0.3, 0.4, 0.0], 0.1)
    assert abs(s.min_distance - 0.4) < 1e-6 and s.position == [s.x, s.y, s.z]
    c = vamp.Cuboid([0.5, 0, 0.5], [0, 0, 0.7], [0.1, 0.2, 0.3])
    assert c.params[11] == 1.0  # yaw only -> axis_3_z == 1 -> filed as z-aligned (environment.cc:123)
    axes = c.params[3:12].reshape(3, 3)
    assert np.allclose(axes @ axes.T, np.eye(3), atol=1e-6)
    c2 = vamp.Cuboid([0.5, 0, 0.5], [0.3, 0.2, 0.7], [0.1, 0.2, 0.3])
    assert c2.params[11] != 1.0
    cyl = vamp.Cylinder([0, 0, 0.5], [0, 0, 0], 0.05, 0.4)  # centre/euler/radius/length
    assert np.allclose(cyl.params[:3], [0, 0, 0.7], atol=1e-6) and np.allclose(cyl.params[3:6], [0, 0, -0.4], atol=1e-6)
    assert np.isclose(cyl.params[7], 1 / 0.16, rtol=1e-6)
    cyl2 = vamp.Cylinder([0, 0, 0], [0.1, 0, 0.2], 0.03)  # endpoints
    assert np.allclose(cyl2.params[3:6], [0.1, 0, 0.2])
    e = vamp.Environment()
    e.add_cuboid(c)
    e.add_cuboid(c2)
    e.add_capsule(cyl)
    e.add_capsule(cyl2)
    e.add_sphere(s)
    t = e.host_tables()
    assert [len(t[k]) for k in ("spheres", "cuboids", "z_cuboids", "capsules", "z_capsules")] == [1, 1, 1, 1, 1]


def test_heightfield_builder_and_oracle_semantics(vamp, oracle):
    """make_heightfield stores reciprocal scales (factory.hh:376-386); the ABI takes the same arguments, keeps at most
    four per environment, and the oracle's lookup follows sphere_heightfield.hh (flat terrain: a sphere collides iff
    its bottom is below the terrain at its centre cell)."""
    import ctypes

    hf = vamp.make_heightfield([0.5, -0.5, 0.1], [0.05, 0.1, 2.0], (4, 3), np.arange(12, dtype=np.float32) / 12)
    assert (hf.x, hf.y, hf.z) == (0.5, -0.5, np.float32(0.1)) and hf.xs == np.float32(1) / np.float32(0.05)
    assert hf.zs == 0.5 and (hf.xd, hf.yd) == (4, 3)
    e = vamp.Environment()
    for _ in range(4):
        e.add_heightfield(hf)
    h = e._build(finalize=False)
    n = ctypes.c_size_t(0)
    assert vamp._lib.lib.vmv_env_heightfield_count(h, ctypes.byref(n)) == 0 and n.value == 4
    vamp._lib.lib.vmv_env_destroy(h)
    e.add_heightfield(hf)
    with pytest.raises(vamp.VmvError) as ei:
        e._build(finalize=False)
    assert ei.value.status == 4  # VMV_ERR_CAPACITY
    with pytest.raises(TypeError):
        vamp.make_heightfield([0, 0, 0], [1, 1, 1], (4, 3), np.zeros(11, np.float32))
    # oracle: flat terrain of height 0.3 (zs * 0.6 + z = 0.5 * 0.6 + 0.0), 1 m x 1 m
    oe = oracle.env()
    oe.add_heightfield([0, 0, 0], [0.1, 0.1, 2.0], 10, 10, np.full(100, 0.6, np.float32))
    f = ctypes.POINTER(ctypes.c_float)
    for c, r, want in (([0.1, 0.1, 0.5], 0.1, 0), ([0.1, 0.1, 0.35], 0.1, 1), ([5.0, -7.0, 0.39], 0.1, 1),
                       ([5.0, -7.0, 0.41], 0.1, 0)):  # far outside the image: clamped border cell, same height
        cc = np.array(c, np.float32)
        assert oracle.L.vo_sphere_environment_in_collision(oe.h, cc.ctypes.data_as(f), ctypes.c_float(r)) == want


def test_compute_fails_loudly_without_gpu(vamp):
    if vamp.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(vamp.VmvError) as ei:
        vamp.panda.validate([0.0] * 7)
    assert ei.value.status == 2  # VMV_ERR_NO_DEVICE: there is no CPU fallback


def test_product_never_touches_the_oracle():
    """The shipped package must not import, link or call anything under oracle/."""
    here = os.path.dirname(os.path.abspath(__file__))
    pkg = os.path.join(here, "..", "vamp_mvt_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".inc", ".cpp")):
                text = open(os.path.join(root, f), errors="replace").read()
                assert "liboracle" not in text and "vamp_oracle" not in text and "oracle_lib" not in text, f


def test_shape_attributes_mirror_the_reference_names(vamp, oracle):
    """read-only parameters of bindings/environment.cc:22-98; min_distance equals the oracle's (collision/shapes.hh)"""
    c = vamp.Cuboid([0.4, -0.2, 0.7], [0.3, -0.5, 1.1], [0.1, 0.05, 0.2])
    assert (c.x, c.y, c.z) == tuple(float(v) for v in c.params[:3]) and c.axis_3_r == float(c.params[14])
    assert abs(c.axis_1_x ** 2 + c.axis_1_y ** 2 + c.axis_1_z ** 2 - 1.0) < 1e-6
    e = oracle.env()
    e.add_cuboid(c.params)
    assert np.float32(c.min_distance) == e.cuboids(False)[0, 15]
    k = vamp.Cylinder([0.4, 0.3, 0.5], [0.2, 0.9, -0.4], 0.06, 0.5)
    assert abs(k.x2 - (k.x1 + k.xv)) < 1e-6 and k.r == float(np.float32(0.06)) and abs(k.rdv * (k.xv ** 2 + k.yv ** 2 + k.zv ** 2) - 1) < 1e-5
    e = oracle.env()
    e.add_capsule(k.params)
    assert np.float32(k.min_distance) == e.capsules(False)[0, 8]
    s = vamp.Sphere([0.3, 0.4, 0.0], 0.1)
    s.name = "ball"
    assert s.name == "ball" and abs(s.min_distance - 0.4) < 1e-6 and s.position == [s.x, s.y, s.z]
