"""ctypes view of oracle/liboracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this.  Nothing under vamp_mvt_amd/ does.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(_HERE, "..", "oracle")
_fp = ctypes.POINTER(ctypes.c_float)
_u8p = ctypes.POINTER(ctypes.c_uint8)


def _f(a):
    return a.ctypes.data_as(_fp)


class MvtView(ctypes.Structure):
    _fields_ = [("grid_width", ctypes.c_uint32), ("capacity", ctypes.c_uint32), ("n_voxels", ctypes.c_uint32),
                ("n_y_tables", ctypes.c_uint32), ("n_z_tables", ctypes.c_uint32),
                ("inverse_scale_factor", ctypes.c_float), ("global_min", ctypes.c_float * 3),
                ("global_max", ctypes.c_float * 3), ("x_table", ctypes.POINTER(ctypes.c_uint32)),
                ("y_tables", ctypes.POINTER(ctypes.c_uint32)), ("z_tables", ctypes.POINTER(ctypes.c_uint32)),
                ("voxel_count", ctypes.POINTER(ctypes.c_uint32)), ("voxel_bbox", _fp), ("px", _fp), ("py", _fp),
                ("pz", _fp)]


class CaptView(ctypes.Structure):
    _fields_ = [("nlog2", ctypes.c_uint32), ("n_tests", ctypes.c_uint32), ("n_leaves", ctypes.c_uint32),
                ("n_aff_vectors", ctypes.c_uint32), ("tests", _fp), ("aff_starts", ctypes.POINTER(ctypes.c_uint32)),
                ("aabbs", _fp), ("aff_x", _fp), ("aff_y", _fp), ("aff_z", _fp), ("aabb_top", ctypes.c_float * 6),
                ("r_min", ctypes.c_float), ("r_max", ctypes.c_float), ("r_point", ctypes.c_float)]


def build_oracle():
    """Compile oracle/liboracle.so if missing or stale (gcc only; no GPU, no reference needed)."""
    so = os.path.join(ORACLE_DIR, "liboracle.so")
    if not all(os.path.exists(os.path.join(ORACLE_DIR, "gen", f)) for f in ("robots_gen.inc", "robots_fk_v8.inc")):
        # generated FK of the oracle (from the committed robot models; needs neither a GPU nor the reference)
        subprocess.check_call([sys.executable, os.path.join(ORACLE_DIR, "..", "tools", "gen_code.py")],
                              stdout=subprocess.DEVNULL)
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("vamp_oracle.c", "vamp_oracle.h", "gen/robots_gen.inc", "gen/robots_fk_v8.inc")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


class Oracle:
    def __init__(self):
        L = ctypes.CDLL(build_oracle())
        self.L = L
        L.vo_env_create.restype = ctypes.c_void_p
        L.vo_env_destroy.argtypes = [ctypes.c_void_p]
        L.vo_env_add_sphere.argtypes = [ctypes.c_void_p] + [ctypes.c_float] * 4
        L.vo_env_add_cuboid.argtypes = [ctypes.c_void_p, _fp]
        L.vo_env_add_capsule.argtypes = [ctypes.c_void_p, _fp]
        L.vo_env_attach.argtypes = [ctypes.c_void_p, _fp, _fp, ctypes.c_size_t]
        L.vo_env_detach.argtypes = [ctypes.c_void_p]
        L.vo_eefk.argtypes = [ctypes.c_int, _fp, _fp]
        L.vo_env_add_heightfield.argtypes = [ctypes.c_void_p, _fp, _fp, ctypes.c_size_t, ctypes.c_size_t, _fp]
        L.vo_env_add_heightfield.restype = ctypes.c_int
        L.vo_env_add_capt.argtypes = [ctypes.c_void_p, _fp, ctypes.c_size_t] + [ctypes.c_float] * 3
        L.vo_env_add_capt.restype = ctypes.c_int
        L.vo_env_add_mvt.argtypes = [ctypes.c_void_p, _fp, ctypes.c_size_t, ctypes.c_float, ctypes.c_float, _fp, _fp,
                                     ctypes.c_float]
        L.vo_env_add_mvt.restype = ctypes.c_int
        L.vo_mvt_collides.argtypes = [ctypes.c_void_p, ctypes.c_size_t, _fp, ctypes.c_float]
        L.vo_mvt_collides_simd.argtypes = [ctypes.c_void_p, ctypes.c_size_t, _fp, _fp, _fp, _fp, ctypes.c_int]
        L.vo_env_mvt_view.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(MvtView)]
        L.vo_env_counts.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t)]
        for fn in ("vo_env_get_spheres",):
            getattr(L, fn).argtypes = [ctypes.c_void_p, _fp]
            getattr(L, fn).restype = ctypes.c_size_t
        for fn in ("vo_env_get_cuboids", "vo_env_get_capsules"):
            getattr(L, fn).argtypes = [ctypes.c_void_p, ctypes.c_int, _fp]
            getattr(L, fn).restype = ctypes.c_size_t
        L.vo_env_capt_view.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(CaptView)]
        L.vo_capt_collides.argtypes = [ctypes.c_void_p, ctypes.c_size_t, _fp, ctypes.c_float]
        L.vo_capt_collides_simd.argtypes = [ctypes.c_void_p, ctypes.c_size_t, _fp, _fp, _fp, _fp, ctypes.c_int]
        L.vo_sphere_environment_in_collision.argtypes = [ctypes.c_void_p, _fp, ctypes.c_float]
        for fn in ("vo_filter_scdf", "vo_filter_centervox"):
            getattr(L, fn).restype = ctypes.c_size_t
        L.vo_filter_scdf.argtypes = [_fp, ctypes.c_size_t, ctypes.c_float, ctypes.c_float, _fp, _fp, _fp, ctypes.c_int, _fp]
        L.vo_filter_centervox.argtypes = [_fp, ctypes.c_size_t, ctypes.c_float, ctypes.c_float, _fp, _fp, _fp, _fp]
        L.vo_robot_id.argtypes = [ctypes.c_char_p]
        for fn in ("vo_robot_dimension", "vo_robot_n_spheres", "vo_robot_n_total_spheres", "vo_robot_resolution"):
            getattr(L, fn).argtypes = [ctypes.c_int]
            getattr(L, fn).restype = ctypes.c_size_t
        L.vo_robot_bounds.argtypes = [ctypes.c_int, _fp, _fp]
        for fn in ("vo_sin", "vo_cos"):
            getattr(L, fn).argtypes = [ctypes.c_float]
            getattr(L, fn).restype = ctypes.c_float
        L.vo_l2_norm.argtypes = [_fp, ctypes.c_size_t]
        L.vo_l2_norm.restype = ctypes.c_float
        L.vo_fk.argtypes = [ctypes.c_int, _fp, _fp]
        L.vo_fk_all.argtypes = [ctypes.c_int, _fp, _fp]
        L.vo_fkcc_rake.argtypes = [ctypes.c_int, ctypes.c_void_p, _fp]
        L.vo_validate.argtypes = [ctypes.c_int, ctypes.c_void_p, _fp, ctypes.c_int]
        L.vo_validate_motion.argtypes = [ctypes.c_int, ctypes.c_void_p, _fp, _fp]
        L.vo_validate_batch.argtypes = [ctypes.c_int, ctypes.c_void_p, _fp, ctypes.c_size_t, _u8p]
        L.vo_validate_motion_batch.argtypes = [ctypes.c_int, ctypes.c_void_p, _fp, _fp, ctypes.c_size_t, _u8p]
        L.vo_validate_batch_mt.argtypes = [ctypes.c_int, ctypes.c_void_p, _fp, ctypes.c_size_t, _u8p, ctypes.c_int]
        L.vo_filter_self_from_pointcloud.argtypes = [ctypes.c_int, ctypes.c_void_p, _fp, _fp, ctypes.c_size_t, ctypes.c_float, _fp]
        L.vo_filter_self_from_pointcloud.restype = ctypes.c_size_t
        L.vo_validate_batch_avx2_mt.argtypes = [ctypes.c_int, ctypes.c_void_p, _fp, ctypes.c_size_t, _u8p, ctypes.c_int]
        L.vo_validate_motion_batch_mt.argtypes = [ctypes.c_int, ctypes.c_void_p, _fp, _fp, ctypes.c_size_t, _u8p, ctypes.c_int]

    def eefk(self, rid, q):
        out = np.zeros((4, 4), np.float32)
        q = np.ascontiguousarray(q, np.float32)
        self.L.vo_eefk(rid, _f(q), _f(out))
        return out

    # -- point-cloud filters ---------------------------------------------------
    def filter_scdf(self, pc, min_dist, max_range, origin, lo, hi, cull):
        pc = np.ascontiguousarray(pc, np.float32).reshape(-1, 3)
        out = np.zeros((max(len(pc), 1), 3), np.float32)
        o, l, h = (np.ascontiguousarray(a, np.float32) for a in (origin, lo, hi))
        m = self.L.vo_filter_scdf(_f(pc), len(pc), min_dist, max_range, _f(o), _f(l), _f(h), int(cull), _f(out))
        return out[:m].copy()

    def filter_centervox(self, pc, voxel_size, max_range, origin, lo, hi):
        """-> points, or None where the reference throws (voxel pool exhausted)"""
        pc = np.ascontiguousarray(pc, np.float32).reshape(-1, 3)
        out = np.zeros((max(len(pc), 1), 3), np.float32)
        o, l, h = (np.ascontiguousarray(a, np.float32) for a in (origin, lo, hi))
        m = self.L.vo_filter_centervox(_f(pc), len(pc), voxel_size, max_range, _f(o), _f(l), _f(h), _f(out))
        return None if m == ctypes.c_size_t(-1).value else out[:m].copy()

    # -- robots -------------------------------------------------------------
    def robot(self, name):
        rid = self.L.vo_robot_id(name.encode())
        if rid < 0:
            raise KeyError(name)
        return rid

    def dimension(self, rid):
        return self.L.vo_robot_dimension(rid)

    def n_spheres(self, rid):
        return self.L.vo_robot_n_spheres(rid)

    def n_total_spheres(self, rid):
        return self.L.vo_robot_n_total_spheres(rid)

    def resolution(self, rid):
        return self.L.vo_robot_resolution(rid)

    def bounds(self, rid):
        d = self.dimension(rid)
        lo, span = np.zeros(d, np.float32), np.zeros(d, np.float32)
        self.L.vo_robot_bounds(rid, _f(lo), _f(span))
        return lo, span

    def sin(self, x):
        return np.array([self.L.vo_sin(float(v)) for v in np.asarray(x, np.float32).ravel()], np.float32)

    def cos(self, x):
        return np.array([self.L.vo_cos(float(v)) for v in np.asarray(x, np.float32).ravel()], np.float32)

    def l2_norm(self, v):
        v = np.ascontiguousarray(v, np.float32)
        return float(self.L.vo_l2_norm(_f(v), v.size))

    def fk(self, rid, q):
        q = np.ascontiguousarray(q, np.float32)
        out = np.zeros((self.n_spheres(rid), 4), np.float32)
        self.L.vo_fk(rid, _f(q), _f(out))
        return out

    def fk_all(self, rid, q):
        q = np.ascontiguousarray(q, np.float32)
        out = np.zeros((self.n_total_spheres(rid), 4), np.float32)
        self.L.vo_fk_all(rid, _f(q), _f(out))
        return out

    # -- environment ----------------------------------------------------------
    def env(self):
        return OracleEnv(self)

    def fkcc_rake(self, rid, env, block):
        """block: [dim][8]"""
        block = np.ascontiguousarray(block, np.float32)
        return bool(self.L.vo_fkcc_rake(rid, env.h, _f(block)))

    def validate(self, rid, env, q, check_bounds=False):
        q = np.ascontiguousarray(q, np.float32)
        return bool(self.L.vo_validate(rid, env.h, _f(q), int(check_bounds)))

    def validate_motion(self, rid, env, a, b):
        a = np.ascontiguousarray(a, np.float32)
        b = np.ascontiguousarray(b, np.float32)
        return bool(self.L.vo_validate_motion(rid, env.h, _f(a), _f(b)))

    def validate_batch(self, rid, env, q, threads=1):
        q = np.ascontiguousarray(q, np.float32)
        out = np.zeros(q.shape[0], np.uint8)
        if threads > 1:
            self.L.vo_validate_batch_mt(rid, env.h, _f(q), q.shape[0], out.ctypes.data_as(_u8p), threads)
        else:
            self.L.vo_validate_batch(rid, env.h, _f(q), q.shape[0], out.ctypes.data_as(_u8p))
        return out.astype(bool)

    def filter_self_from_pointcloud(self, rid, env, q, pts, r):
        q = np.ascontiguousarray(q, np.float32)
        pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 3)
        out = np.zeros((max(len(pts), 1), 3), np.float32)
        m = self.L.vo_filter_self_from_pointcloud(rid, env.h, _f(q), _f(pts), len(pts), float(r), _f(out))
        return out[:m].copy()

    def has_avx2(self):
        return bool(self.L.vo_has_avx2())

    def validate_batch_avx2(self, rid, env, q, threads=1):
        """the AVX2 rake-of-8 build of validate_batch (primitive environments only)"""
        q = np.ascontiguousarray(q, np.float32)
        out = np.zeros(q.shape[0], np.uint8)
        if self.L.vo_validate_batch_avx2_mt(rid, env.h, _f(q), q.shape[0], out.ctypes.data_as(_u8p), threads) != 0:
            raise ValueError("environment not covered by the AVX2 build (or no AVX2)")
        return out.astype(bool)

    def validate_motion_batch(self, rid, env, a, b, threads=1):
        a = np.ascontiguousarray(a, np.float32)
        b = np.ascontiguousarray(b, np.float32)
        out = np.zeros(a.shape[0], np.uint8)
        if threads > 1:
            self.L.vo_validate_motion_batch_mt(rid, env.h, _f(a), _f(b), a.shape[0], out.ctypes.data_as(_u8p), threads)
        else:
            self.L.vo_validate_motion_batch(rid, env.h, _f(a), _f(b), a.shape[0], out.ctypes.data_as(_u8p))
        return out.astype(bool)


class OracleEnv:
    def __init__(self, oracle):
        self.o = oracle
        self.h = ctypes.c_void_p(oracle.L.vo_env_create())

    def __del__(self):
        try:
            self.o.L.vo_env_destroy(self.h)
        except Exception:
            pass

    def add_sphere(self, x, y, z, r):
        self.o.L.vo_env_add_sphere(self.h, float(x), float(y), float(z), float(r))

    def add_cuboid(self, p15):
        p = np.ascontiguousarray(p15, np.float32)
        assert p.size == 15
        self.o.L.vo_env_add_cuboid(self.h, _f(p))

    def attach(self, tf, spheres):
        tf = np.ascontiguousarray(tf, np.float32).reshape(4, 4)
        sp = np.ascontiguousarray(spheres, np.float32).reshape(-1, 4)
        self.o.L.vo_env_attach(self.h, _f(tf), _f(sp), sp.shape[0])

    def detach(self):
        self.o.L.vo_env_detach(self.h)

    def add_heightfield(self, center, scale, xd, yd, data):
        c, sc, d = (np.ascontiguousarray(a, np.float32) for a in (center, scale, data))
        assert d.size == xd * yd
        assert self.o.L.vo_env_add_heightfield(self.h, _f(c), _f(sc), xd, yd, _f(d.reshape(-1))) == 0

    def add_capsule(self, p8):
        p = np.ascontiguousarray(p8, np.float32)
        assert p.size == 8
        self.o.L.vo_env_add_capsule(self.h, _f(p))

    def add_capt(self, points, r_min, r_max, r_point):
        p = np.ascontiguousarray(points, np.float32)
        rc = self.o.L.vo_env_add_capt(self.h, _f(p), p.shape[0], float(r_min), float(r_max), float(r_point))
        if rc != 0:
            raise ValueError("capt build failed")

    def add_mvt(self, points, r_min, r_max, ws_min, ws_max, r_point):
        """-> 0 ok, else the reason code under which the reference would throw"""
        p = np.ascontiguousarray(points, np.float32)
        lo, hi = np.ascontiguousarray(ws_min, np.float32), np.ascontiguousarray(ws_max, np.float32)
        return int(self.o.L.vo_env_add_mvt(self.h, _f(p), p.shape[0], float(r_min), float(r_max), _f(lo), _f(hi),
                                           float(r_point)))

    def mvt(self, index=0):
        v = MvtView()
        if self.o.L.vo_env_mvt_view(self.h, index, ctypes.byref(v)) != 0:
            raise IndexError(index)
        return dict(grid_width=v.grid_width, capacity=v.capacity, n_voxels=v.n_voxels, n_z_tables=v.n_z_tables,
                    inverse_scale_factor=v.inverse_scale_factor,
                    global_box=np.array(list(v.global_min) + list(v.global_max), np.float32))

    def mvt_collides(self, c, r, index=0):
        c = np.ascontiguousarray(c, np.float32)
        return bool(self.o.L.vo_mvt_collides(self.h, index, _f(c), float(r)))

    def mvt_collides_simd(self, cx, cy, cz, r, index=0):
        cx, cy, cz, r = (np.ascontiguousarray(a, np.float32) for a in (cx, cy, cz, r))
        return bool(self.o.L.vo_mvt_collides_simd(self.h, index, _f(cx), _f(cy), _f(cz), _f(r), cx.size))

    def counts(self):
        c = (ctypes.c_size_t * 6)()
        self.o.L.vo_env_counts(self.h, c)
        return list(c)

    def spheres(self):
        n = self.counts()[0]
        out = np.zeros((max(n, 1), 5), np.float32)
        self.o.L.vo_env_get_spheres(self.h, _f(out))
        return out[:n]

    def cuboids(self, z_aligned):
        n = self.counts()[4 if z_aligned else 3]
        out = np.zeros((max(n, 1), 16), np.float32)
        self.o.L.vo_env_get_cuboids(self.h, int(z_aligned), _f(out))
        return out[:n]

    def capsules(self, z_aligned):
        n = self.counts()[2 if z_aligned else 1]
        out = np.zeros((max(n, 1), 9), np.float32)
        self.o.L.vo_env_get_capsules(self.h, int(z_aligned), _f(out))
        return out[:n]

    def capt(self, index=0):
        v = CaptView()
        if self.o.L.vo_env_capt_view(self.h, index, ctypes.byref(v)) != 0:
            raise IndexError(index)
        n_l, n_a = v.n_leaves, v.n_aff_vectors
        return dict(
            nlog2=v.nlog2,
            tests=np.ctypeslib.as_array(v.tests, (v.n_tests,)).copy(),
            aff_starts=np.ctypeslib.as_array(v.aff_starts, (n_l + 1,)).copy(),
            aabbs=np.ctypeslib.as_array(v.aabbs, (n_l, 6)).copy(),
            aff=np.stack([np.ctypeslib.as_array(p, (n_a, 8)).copy() for p in (v.aff_x, v.aff_y, v.aff_z)]),
            aabb_top=np.array(list(v.aabb_top), np.float32), r_min=v.r_min, r_max=v.r_max, r_point=v.r_point)

    def capt_collides(self, c, r, index=0):
        c = np.ascontiguousarray(c, np.float32)
        return bool(self.o.L.vo_capt_collides(self.h, index, _f(c), float(r)))

    def capt_collides_simd(self, cx, cy, cz, r, index=0):
        cx, cy, cz, r = (np.ascontiguousarray(a, np.float32) for a in (cx, cy, cz, r))
        return bool(self.o.L.vo_capt_collides_simd(self.h, index, _f(cx), _f(cy), _f(cz), _f(r), cx.size))


SPHERE_CAGE = [  # reference scripts/sphere_cage_example.py:16-31 (problem data), radius 0.2
    [0.55, 0, 0.25], [0.35, 0.35, 0.25], [0, 0.55, 0.25], [-0.55, 0, 0.25], [-0.35, -0.35, 0.25], [0, -0.55, 0.25],
    [0.35, -0.35, 0.25], [0.35, 0.35, 0.8], [0, 0.55, 0.8], [-0.35, 0.35, 0.8], [-0.55, 0, 0.8], [-0.35, -0.35, 0.8],
    [0, -0.55, 0.8], [0.35, -0.35, 0.8]]
CAGE_START = [0., -0.785, 0., -2.356, 0., 1.571, 0.785]
CAGE_GOAL = [2.35, 1., 0., -0.8, 0, 2.5, 0.785]
