"""vamp_mvt_amd — MI355X-native motion validation behind the `vamp` names for this path.

Host-side mirror of the reference's Python surface for ONE hot path (reference `vamp._core`,
src/impl/vamp/bindings/{environment.cc,robot_helper.hh}): `Sphere`, `Cuboid`, `Cylinder`, `Environment`
and the per-robot modules `panda`, `ur5`, `fetch`, `baxter` with `validate`, `fk`, `dimension`,
`resolution`, `n_spheres`, `min_max_radii`, `joint_names`, `end_effector` — plus the batched calls the
reference lists as planned (`README.md:346`): `validate_batch`, `validate_motion_batch`, `fk_batch`.

Everything computes on the GPU through the C ABI (include/vamp_mvt_amd.h -> libvamp_mvt_amd.so).  There is
no CPU path: without a HIP device every compute call raises VmvError(VMV_ERR_NO_DEVICE).
"""
from __future__ import annotations

import ctypes
import math
import sys
import types

import numpy as np

from . import _lib
from ._lib import VmvError, check, lib

__all__ = ["robots", "Sphere", "Cuboid", "Cylinder", "Attachment", "filter_pointcloud", "HeightField", "make_heightfield", "png_to_heightfield", "Environment", "device_count", "set_device", "abi_version",
           "VmvError", "unpack_bits", "POINT_RADIUS"]

POINT_RADIUS = 0.0025  # reference src/vamp/constants.py:25


def abi_version() -> int:
    return lib.vmv_abi_version()


def device_count() -> int:
    n = ctypes.c_int(0)
    lib.vmv_device_count(ctypes.byref(n))
    return n.value


def set_device(index: int) -> None:
    check(lib.vmv_set_device(int(index)), "vmv_set_device")


class _RobotList(list):
    """`vamp.robots` is a list in the reference package (src/vamp/__init__.py:46) and `robots()` a function in its
    extension module (bindings/python.cc.in); this is both"""

    def __call__(self):
        return list(self)


robots = _RobotList(lib.vmv_robot_name(i).decode() for i in range(lib.vmv_num_robots()))


def _f32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise TypeError(f"expected float32 array of shape {shape}, got {a.shape}")
    return a


def _fp(a):
    return a.ctypes.data_as(_lib.c_float_p)


def unpack_bits(words: np.ndarray, n: int) -> np.ndarray:
    """uint64 validity words (bit i%64 of word i//64) -> bool[n]."""
    b = np.unpackbits(np.ascontiguousarray(words, dtype="<u8").view(np.uint8), bitorder="little")
    return b[:n].astype(bool)


# ---------------------------------------------------------------------------------------------------------
# shapes (reference collision/shapes.hh, collision/factory.hh; Python classes of bindings/environment.cc:22-99)
# ---------------------------------------------------------------------------------------------------------
def _rotation_zyx(rho, theta, phi):
    """AngleAxis(phi, Z) * AngleAxis(theta, Y) * AngleAxis(rho, X) in fp32 (collision/factory.hh:37-39).

    Parity note: the reference evaluates this with Eigen's float AngleAxis/matrix products; Eigen is not
    available offline, so the result is equal to tolerance, not pinned bit-for-bit ("parity unpinned").
    Callers that need bit-exact primitives pass canonical parameters (`Cuboid.from_canonical`)."""
    f = np.float32
    cr, sr = f(math.cos(f(rho))), f(math.sin(f(rho)))
    ct, st = f(math.cos(f(theta))), f(math.sin(f(theta)))
    cp, sp = f(math.cos(f(phi))), f(math.sin(f(phi)))
    rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]], np.float32)
    ry = np.array([[ct, 0, st], [0, 1, 0], [-st, 0, ct]], np.float32)
    rz = np.array([[cp, -sp, 0], [sp, cp, 0], [0, 0, 1]], np.float32)
    return (rz @ ry @ rx).astype(np.float32)


class Sphere:
    """vamp.Sphere(center, radius) — bindings/environment.cc:22-40."""

    def __init__(self, center, radius):
        c = _f32(center, (3,))
        self.x, self.y, self.z, self.r = float(c[0]), float(c[1]), float(c[2]), float(np.float32(radius))
        self.name = ""

    @property
    def position(self):
        return [self.x, self.y, self.z]

    @property
    def min_distance(self):
        f = np.float32
        return float(np.sqrt(f(self.x) * f(self.x) + f(self.y) * f(self.y) + f(self.z) * f(self.z)) - f(self.r))

    def __repr__(self):
        return f"Sphere(({self.x}, {self.y}, {self.z}), r={self.r})"


class Cuboid:
    """vamp.Cuboid(center, euler_xyz, half_extents) — collision/factory.hh:26-101."""

    def __init__(self, center, euler_xyz, half_extents):
        c, e, h = _f32(center, (3,)), _f32(euler_xyz, (3,)), _f32(half_extents, (3,))
        rot = _rotation_zyx(e[0], e[1], e[2])
        self.params = np.concatenate([c, rot[:, 0], rot[:, 1], rot[:, 2], h]).astype(np.float32)
        self.name = ""

    @classmethod
    def from_canonical(cls, params15):
        """centre xyz | axis_1 xyz | axis_2 xyz | axis_3 xyz | half extents (collision/shapes.hh:32-49)."""
        self = cls.__new__(cls)
        self.params = _f32(params15, (15,)).copy()
        self.name = ""
        return self

    x = property(lambda s: float(s.params[0]))
    y = property(lambda s: float(s.params[1]))
    z = property(lambda s: float(s.params[2]))
    # read-only parameters under the reference's names (bindings/environment.cc:60-98)
    axis_1_x, axis_1_y, axis_1_z = (property(lambda s, i=i: float(s.params[i])) for i in (3, 4, 5))
    axis_2_x, axis_2_y, axis_2_z = (property(lambda s, i=i: float(s.params[i])) for i in (6, 7, 8))
    axis_3_x, axis_3_y, axis_3_z = (property(lambda s, i=i: float(s.params[i])) for i in (9, 10, 11))
    axis_1_r, axis_2_r, axis_3_r = (property(lambda s, i=i: float(s.params[i])) for i in (12, 13, 14))

    @property
    def min_distance(self):
        """collision/shapes.hh:52-67 (computed by the C ABI's host code, the same function the kernels' tables use)"""
        e = Environment()
        e.add_cuboid(self)
        t = e.host_tables()
        return float((t["cuboids"] if len(t["cuboids"]) else t["z_cuboids"])[0, 15])


class Attachment:
    """vamp.Attachment(tf) — bindings/environment.cc:241-269, collision/attachments.hh: spheres rigidly attached at a
    frame `tf` (4 x 4) relative to the end effector.  add_sphere / add_spheres, relative_frame, set_ee_pose(tf) +
    posed_spheres (host-side float arithmetic, for inspection; the kernels pose per configuration themselves)."""

    def __init__(self, tf):
        self.relative_frame = _f32(tf, (4, 4)).copy()
        self.spheres = []
        self.posed_spheres = []

    def add_sphere(self, sphere: "Sphere"):
        self.spheres.append(sphere)

    def add_spheres(self, spheres):
        self.spheres.extend(spheres)

    def set_ee_pose(self, tf):
        n_tf = (_f32(tf, (4, 4)).astype(np.float64) @ self.relative_frame.astype(np.float64))
        self.posed_spheres = [Sphere((n_tf[:3, :3] @ np.array([s.x, s.y, s.z]) + n_tf[:3, 3]).astype(np.float32), s.r)
                              for s in self.spheres]

    def _arrays(self):
        sp = np.array([[s.x, s.y, s.z, s.r] for s in self.spheres], np.float32).reshape(-1, 4)
        return self.relative_frame.copy(), sp


def filter_pointcloud(pc, min_dist, max_range, voxel_size, origin, workcell_min, workcell_max, cull, filter_type,
                      return_device_time=False):
    """vamp.filter_pointcloud (bindings/environment.cc:183-239) -> (points [m][3] float32, nanoseconds).
    filter_type "scdf" = collision/filter.hh:175-275, "centervox" = collision/filter_centervox.hh; both run on the GPU
    and return the reference's points in the reference's order.  (The reference returns an empty list for any other
    filter_type; this raises, as its Python caller src/vamp/pointcloud.py:145 does.)"""
    if filter_type not in ("scdf", "centervox"):
        raise ValueError("filter_type must be one of: 'scdf', 'centervox'")
    pts = _f32(pc).reshape(-1, 3)
    n = pts.shape[0]
    out = np.zeros((max(n, 1), 3), np.float32)
    m, ns, dns = ctypes.c_size_t(0), ctypes.c_uint64(0), ctypes.c_uint64(0)
    check(lib.vmv_filter_pointcloud(_fp(pts), n, float(min_dist), float(max_range), float(voxel_size), _fp(_f32(origin)),
                                    _fp(_f32(workcell_min)), _fp(_f32(workcell_max)), int(bool(cull)),
                                    0 if filter_type == "scdf" else 1, _fp(out), n, ctypes.byref(m), ctypes.byref(ns),
                                    ctypes.byref(dns)), "vmv_filter_pointcloud")
    res = out[:m.value].copy()
    return (res, int(ns.value), int(dns.value)) if return_device_time else (res, int(ns.value))


class HeightField:
    """vamp.HeightField (bindings/environment.cc:102-109, collision/shapes.hh:250-312): flattened `data` with
    xd = dimensions[0] used as the row length of the lookup (index = ys * xd + xs, sphere_heightfield.hh:22), exactly
    as the reference does.  xs, ys, zs are the reciprocal scales (collision/factory.hh:376-386)."""

    def __init__(self, center, scaling, dimensions, data):
        self.center = _f32(center).reshape(3).copy()
        self.scaling = _f32(scaling).reshape(3).copy()
        self.xd, self.yd = int(dimensions[0]), int(dimensions[1])
        self.data = _f32(data).reshape(-1).copy()
        if self.data.size != self.xd * self.yd:
            raise TypeError("data must hold dimensions[0] * dimensions[1] values")
        self.x, self.y, self.z = (float(v) for v in self.center)
        one = np.float32(1.0)
        self.xs, self.ys, self.zs = (float(one / v) for v in self.scaling)


def make_heightfield(center, scaling, dimensions, data) -> HeightField:
    """vamp.make_heightfield (bindings/environment.cc:100)."""
    return HeightField(center, scaling, dimensions, data)


def png_to_heightfield(filename, center, scaling) -> HeightField:
    """vamp.png_to_heightfield (reference src/vamp/__init__.py:54-66): grey-scale image / 255, flipped vertically;
    dimensions = array.shape, as the reference passes them."""
    from PIL import Image  # same optional dependency as the reference

    array = np.asarray(Image.open(filename).convert("L")) * 1 / 255.0
    array = np.flip(array, axis=0)
    return make_heightfield(center, scaling, array.shape, list(array.flatten()))


class Cylinder:
    """vamp.Cylinder(center, euler_xyz, radius, length) | Cylinder(endpoint1, endpoint2, radius)
    — collision/factory.hh:104-223; the environment treats it as a capsule (environment.cc:134-147)."""

    def __init__(self, *args):
        f = np.float32
        if len(args) == 4:
            c, e = _f32(args[0], (3,)), _f32(args[1], (3,))
            radius, length = f(args[2]), f(args[3])
            rot = _rotation_zyx(e[0], e[1], e[2])
            half = f(length / f(2))
            p1 = (c + rot[:, 2] * half).astype(np.float32)
            p2 = (c - rot[:, 2] * half).astype(np.float32)
        elif len(args) == 3:
            p1, p2, radius = _f32(args[0], (3,)), _f32(args[1], (3,)), f(args[2])
        else:
            raise TypeError("Cylinder(center, euler_xyz, radius, length) or Cylinder(endpoint1, endpoint2, radius)")
        v = (p2 - p1).astype(np.float32)
        dot = f(f(v[0] * v[0]) + f(v[1] * v[1])) + f(v[2] * v[2])
        rdv = f(1.0 / float(dot))  # static_cast<float>(1.0 / dot): double division, then narrowed
        self.params = np.array([p1[0], p1[1], p1[2], v[0], v[1], v[2], radius, rdv], np.float32)
        self.name = ""

    # read-only parameters under the reference's names (bindings/environment.cc:42-58)
    x1, y1, z1, xv, yv, zv, r, rdv = (property(lambda s, i=i: float(s.params[i])) for i in range(8))
    x2 = property(lambda s: float(np.float32(s.params[0]) + np.float32(s.params[3])))
    y2 = property(lambda s: float(np.float32(s.params[1]) + np.float32(s.params[4])))
    z2 = property(lambda s: float(np.float32(s.params[2]) + np.float32(s.params[5])))

    @property
    def min_distance(self):
        """collision/shapes.hh:165-189"""
        e = Environment()
        e.add_capsule(self)
        t = e.host_tables()
        return float((t["capsules"] if len(t["capsules"]) else t["z_capsules"])[0, 8])

    @classmethod
    def from_canonical(cls, params8):
        """x1 y1 z1 | xv yv zv | r | rdv (collision/shapes.hh:128-143)."""
        self = cls.__new__(cls)
        self.params = _f32(params8, (8,)).copy()
        self.name = ""
        return self


class Environment:
    """vamp.Environment — bindings/environment.cc:111-163 (write-only from Python there, too).

    Host-side this is a recipe; the device image is (re)built lazily by the C ABI on first use after a change
    (the reference re-sorts on every add and converts the whole environment on every call)."""

    def __init__(self):
        self._ops = []
        self._handle = None
        self._device = None

    # -- mutation ------------------------------------------------------------------------------------------
    def _dirty(self):
        if self._handle is not None:
            lib.vmv_env_destroy(self._handle)
            self._handle = None

    def add_sphere(self, sphere: Sphere):
        self._ops.append(("sphere", (sphere.x, sphere.y, sphere.z, sphere.r)))
        self._dirty()

    def add_cuboid(self, cuboid: Cuboid):
        self._ops.append(("cuboid", cuboid.params.copy()))
        self._dirty()

    def add_capsule(self, capsule: Cylinder):
        self._ops.append(("capsule", capsule.params.copy()))
        self._dirty()

    def attach(self, attachment: Attachment):
        """Environment.attach (environment.cc:178-180): at most one attachment; a later one replaces it."""
        self._ops = [op for op in self._ops if op[0] != "attach"]
        self._ops.append(("attach", attachment._arrays()))
        self._dirty()

    def detach(self):
        self._ops = [op for op in self._ops if op[0] != "attach"]
        self._dirty()

    def add_heightfield(self, heightfield: HeightField):
        self._ops.append(("heightfield", heightfield))
        self._dirty()

    def add_capt_pointcloud(self, points, r_min, r_max, r_point, build="host", return_device_time=False):
        """-> CAPT build time in nanoseconds (environment.cc:152-163).  build="gpu" builds the same arrays on the
        device (csrc/vmv_capt_gpu.hip); with return_device_time also the HIP-event time of the device work alone."""
        if build not in ("host", "gpu"):
            raise ValueError("build must be 'host' or 'gpu'")
        pts = _f32(points)
        if pts.ndim != 2 or pts.shape[1] != 3:
            raise TypeError("points must be [n][3]")
        self._ops.append(("capt_gpu" if build == "gpu" else "capt", (pts.copy(), float(r_min), float(r_max), float(r_point))))
        self._dirty()
        # build once now to report the time, as the reference does
        h = ctypes.c_void_p()
        check(lib.vmv_env_create(ctypes.byref(h)), "vmv_env_create")
        ns, dns = ctypes.c_uint64(0), ctypes.c_uint64(0)
        try:
            if build == "gpu":
                check(lib.vmv_env_add_capt_pointcloud_gpu(h, _fp(pts), pts.shape[0], r_min, r_max, r_point, ctypes.byref(ns),
                                                          ctypes.byref(dns)), "vmv_env_add_capt_pointcloud_gpu")
            else:
                check(lib.vmv_env_add_capt_pointcloud(h, _fp(pts), pts.shape[0], r_min, r_max, r_point, ctypes.byref(ns)),
                      "vmv_env_add_capt_pointcloud")
        finally:
            lib.vmv_env_destroy(h)
        return (int(ns.value), int(dns.value)) if return_device_time else int(ns.value)

    def add_mvt_pointcloud(self, points, r_min, r_max, workspace_aabb_min, workspace_aabb_max, r_point):
        """-> MVT build time in nanoseconds (environment.cc:164-177).  Raises VmvError(VMV_ERR_CAPACITY) where the
        reference would terminate on an exhausted pool (mvt.hh:634-648)."""
        pts = _f32(points)
        if pts.ndim != 2 or pts.shape[1] != 3:
            raise TypeError("points must be [n][3]")
        lo, hi = _f32(workspace_aabb_min, (3,)), _f32(workspace_aabb_max, (3,))
        h = ctypes.c_void_p()
        check(lib.vmv_env_create(ctypes.byref(h)), "vmv_env_create")
        ns = ctypes.c_uint64(0)
        try:
            check(lib.vmv_env_add_mvt_pointcloud(h, _fp(pts), pts.shape[0], r_min, r_max, _fp(lo), _fp(hi), r_point,
                                                 ctypes.byref(ns), None), "vmv_env_add_mvt_pointcloud")
        finally:
            lib.vmv_env_destroy(h)
        self._ops.append(("mvt", (pts.copy(), float(r_min), float(r_max), lo.copy(), hi.copy(), float(r_point))))
        self._dirty()
        return int(ns.value)

    def __del__(self):
        try:
            self._dirty()
        except Exception:
            pass

    # -- device image ----------------------------------------------------------------------------------------
    def _build(self, finalize=True):
        h = ctypes.c_void_p()
        check(lib.vmv_env_create(ctypes.byref(h)), "vmv_env_create")
        try:
            for kind, arg in self._ops:
                if kind == "sphere":
                    check(lib.vmv_env_add_sphere(h, *arg), "vmv_env_add_sphere")
                elif kind == "cuboid":
                    check(lib.vmv_env_add_cuboid(h, _fp(arg)), "vmv_env_add_cuboid")
                elif kind == "capsule":
                    check(lib.vmv_env_add_capsule(h, _fp(arg)), "vmv_env_add_capsule")
                elif kind == "attach":
                    tf, sp = arg
                    check(lib.vmv_env_attach(h, _fp(tf), _fp(sp), sp.shape[0]), "vmv_env_attach")
                elif kind == "heightfield":
                    check(lib.vmv_env_add_heightfield(h, _fp(arg.center), _fp(arg.scaling), arg.xd, arg.yd, _fp(arg.data)),
                          "vmv_env_add_heightfield")
                elif kind == "mvt":
                    pts, r_min, r_max, lo, hi, r_point = arg
                    check(lib.vmv_env_add_mvt_pointcloud(h, _fp(pts), pts.shape[0], r_min, r_max, _fp(lo), _fp(hi),
                                                         r_point, None, None), "vmv_env_add_mvt_pointcloud")
                elif kind == "capt_gpu":
                    pts, r_min, r_max, r_point = arg
                    check(lib.vmv_env_add_capt_pointcloud_gpu(h, _fp(pts), pts.shape[0], r_min, r_max, r_point, None, None),
                          "vmv_env_add_capt_pointcloud_gpu")
                else:
                    pts, r_min, r_max, r_point = arg
                    check(lib.vmv_env_add_capt_pointcloud(h, _fp(pts), pts.shape[0], r_min, r_max, r_point, None),
                          "vmv_env_add_capt_pointcloud")
            if finalize:
                check(lib.vmv_env_finalize(h), "vmv_env_finalize")
        except Exception:
            lib.vmv_env_destroy(h)
            raise
        return h

    def handle(self):
        """Finalized C-ABI handle (sorted + uploaded to the current device; rebuilt if the device changed)."""
        dev = ctypes.c_int(-1)
        lib.vmv_get_device(ctypes.byref(dev))
        if self._handle is not None and self._device != dev.value:
            self._dirty()
        if self._handle is None:
            self._handle = self._build()
            self._device = dev.value
        return self._handle

    def spheres_in_collision(self, spheres):
        """sphere_environment_in_collision (collision/validity.hh:47-158) for free spheres [n][4] = x y z r -> bool[n]
        (each sphere on its own: primitive lists, heightfields, CAPT and MVT clouds)."""
        s = _f32(spheres)
        if s.ndim != 2 or s.shape[1] != 4:
            raise TypeError("spheres must be [n][4] = x y z r")
        hits = np.zeros(s.shape[0], np.uint8)
        check(lib.vmv_spheres_in_collision_batch_host(self.handle(), _fp(s), s.shape[0],
                                                      hits.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))),
              "vmv_spheres_in_collision_batch_host")
        return hits.astype(bool)

    # -- inspection (host tables; no GPU needed) ----------------------------------------------------------------
    def host_tables(self):
        """Sorted primitive tables as the kernels see them (built without uploading)."""
        h = self._build(finalize=False)
        try:
            counts = (ctypes.c_size_t * 6)()
            check(lib.vmv_env_counts(h, counts), "vmv_env_counts")
            n = ctypes.c_size_t(0)
            out = {}
            sp = np.zeros((max(counts[0], 1), 5), np.float32)
            lib.vmv_env_get_spheres(h, _fp(sp), counts[0], ctypes.byref(n))
            out["spheres"] = sp[:counts[0]]
            for key, z, cnt, width, fn in (("cuboids", 0, counts[3], 16, lib.vmv_env_get_cuboids),
                                           ("z_cuboids", 1, counts[4], 16, lib.vmv_env_get_cuboids),
                                           ("capsules", 0, counts[1], 9, lib.vmv_env_get_capsules),
                                           ("z_capsules", 1, counts[2], 9, lib.vmv_env_get_capsules)):
                a = np.zeros((max(cnt, 1), width), np.float32)
                fn(h, z, _fp(a), cnt, ctypes.byref(n))
                out[key] = a[:cnt]
            capts = []
            for i in range(counts[5]):
                nlog2, naff = ctypes.c_uint32(0), ctypes.c_uint32(0)
                check(lib.vmv_env_capt_sizes(h, i, ctypes.byref(nlog2), ctypes.byref(naff)), "vmv_env_capt_sizes")
                leaves = 1 << nlog2.value
                t = np.zeros(leaves - 1, np.float32)
                st = np.zeros(leaves + 1, np.uint32)
                bb = np.zeros((leaves, 6), np.float32)
                aff = np.zeros((3, naff.value, 8), np.float32)
                top = np.zeros(6, np.float32)
                check(lib.vmv_env_capt_arrays(h, i, _fp(t), st.ctypes.data_as(_lib.c_u32_p), _fp(bb), _fp(aff[0]),
                                              _fp(aff[1]), _fp(aff[2]), _fp(top)), "vmv_env_capt_arrays")
                capts.append(dict(nlog2=nlog2.value, tests=t, aff_starts=st, aabbs=bb, aff=aff, aabb_top=top))
            out["capt"] = capts
            return out
        finally:
            lib.vmv_env_destroy(h)


# ---------------------------------------------------------------------------------------------------------
# per-robot modules (reference bindings/robot_helper.hh:326-597, path subset + batched calls)
# ---------------------------------------------------------------------------------------------------------
def _is_torch_cuda(x):
    t = sys.modules.get("torch")
    return t is not None and isinstance(x, t.Tensor) and x.is_cuda


class _Robot(types.ModuleType):
    def __init__(self, name: str):
        super().__init__(f"{__name__}.{name}")
        self._id = lib.vmv_robot_id(name.encode())
        if self._id < 0:
            raise ImportError(f"robot {name} missing from the library")
        self._name = name
        self._dim = lib.vmv_robot_dimension(self._id)
        self._ns = lib.vmv_robot_n_spheres(self._id)
        lo, sp, ds = (np.zeros(self._dim, np.float32) for _ in range(3))
        check(lib.vmv_robot_bounds(self._id, _fp(lo), _fp(sp), _fp(ds)), "vmv_robot_bounds")
        self._lower, self._span, self._descale = lo, sp, ds

    # constants ---------------------------------------------------------------------------------------------
    def dimension(self):
        return self._dim

    def resolution(self):
        return lib.vmv_robot_resolution(self._id)

    def n_spheres(self):
        return self._ns

    def min_max_radii(self):
        a, b = ctypes.c_float(0), ctypes.c_float(0)
        check(lib.vmv_robot_min_max_radii(self._id, ctypes.byref(a), ctypes.byref(b)), "vmv_robot_min_max_radii")
        return (a.value, b.value)

    def joint_names(self):
        return [lib.vmv_robot_joint_name(self._id, j).decode() for j in range(self._dim)]

    def end_effector(self):
        return lib.vmv_robot_end_effector(self._id).decode()

    def lower_bounds(self):
        return self._lower.copy()

    def upper_bounds(self):
        return (self._lower + self._span).astype(np.float32)

    # single-configuration calls (drop-in names) --------------------------------------------------------------
    def validate(self, configuration, environment: Environment | None = None, check_bounds: bool = False) -> bool:
        """<robot>.validate(q, env=Environment(), check_bounds=False) — robot_helper.hh:255-267."""
        q = _f32(configuration, (self._dim,))
        if check_bounds:
            t = ((q - self._lower) * self._descale).astype(np.float32)  # descale_configuration, panda.hh:82-85
            if not (bool((t <= np.float32(1)).all()) and bool((t >= np.float32(0)).all())):
                return False
        return bool(self.validate_batch(q[None, :], environment)[0])

    def validate_motion(self, start, goal, environment: Environment | None = None) -> bool:
        """validate_motion<Robot, 8, resolution>(start, goal, env) — planning/validate.hh:70-77."""
        a, b = _f32(start, (self._dim,)), _f32(goal, (self._dim,))
        return bool(self.validate_motion_batch(a[None, :], b[None, :], environment)[0])

    def debug(self, configuration, environment: Environment | None = None):
        """<robot>.debug(q, env) — robot_helper.hh:249-253 (Robot::fkcc_debug): (per fine sphere the objects it collides
        with, the fine sphere pairs of the self-collision groups that overlap).  Objects are reported as
        (list, position) = position in the environment's sorted "spheres" / "capsules" / "z_capsules" / "cuboids" /
        "z_cuboids" tables (Environment.host_tables()) or ("heightfield", index) — the reference reports `name`s, which
        this environment does not keep."""
        q = _f32(configuration, (self._dim,))[None, :]
        h = self._env(environment)
        ns = self.n_spheres()
        words = np.zeros((1, ns, 9), np.uint32)
        npairs = ctypes.c_size_t(0)
        check(lib.vmv_robot_self_pairs(self._id, ctypes.byref(npairs), None), "vmv_robot_self_pairs")
        pairs = np.zeros((max(npairs.value, 1), 2), np.uint16)
        check(lib.vmv_robot_self_pairs(self._id, ctypes.byref(npairs), pairs.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16))),
              "vmv_robot_self_pairs")
        pw = np.zeros((1, max((npairs.value + 31) // 32, 1)), np.uint32)
        check(lib.vmv_contacts_batch_host(self._id, h, _fp(q), 1, words.ctypes.data_as(_lib.c_u32_p),
                                          pw.ctypes.data_as(_lib.c_u32_p)), "vmv_contacts_batch_host")
        base = np.zeros(5, np.uint32)
        check(lib.vmv_env_report_layout(h, base.ctypes.data_as(_lib.c_u32_p)), "vmv_env_report_layout")
        counts = (ctypes.c_size_t * 6)()
        check(lib.vmv_env_counts(h, counts), "vmv_env_counts")
        names = ("spheres", "capsules", "z_capsules", "cuboids", "z_cuboids")
        per_sphere = []
        for s in range(ns):
            hits = []
            for li, name in enumerate(names):
                for i in range(counts[li]):
                    if (int(words[0, s, int(base[li]) + i // 32]) >> (i % 32)) & 1:
                        hits.append((name, i))
            hits += [("heightfield", i) for i in range(4) if (int(words[0, s, 8]) >> i) & 1]
            per_sphere.append(hits)
        hit_pairs = [(int(pairs[p, 0]), int(pairs[p, 1])) for p in range(npairs.value)
                     if (int(pw[0, p // 32]) >> (p % 32)) & 1]
        return per_sphere, hit_pairs

    def eefk(self, configuration):
        """<robot>.eefk(q) -> 4 x 4 end-effector frame — robot_helper.hh:279-282."""
        return self.eefk_batch(_f32(configuration, (self._dim,))[None, :])[0]

    def eefk_batch(self, configurations):
        q = _f32(configurations)
        if q.ndim != 2 or q.shape[1] != self._dim:
            raise TypeError(f"expected [n][{self._dim}] configurations")
        out = np.zeros((q.shape[0], 4, 4), np.float32)
        check(lib.vmv_eefk_batch_host(self._id, _fp(q), q.shape[0], _fp(out)), "vmv_eefk_batch_host")
        return out

    def fk(self, configuration):
        """<robot>.fk(q) -> list[Sphere] — robot_helper.hh:234-247."""
        out = self.fk_batch(_f32(configuration, (self._dim,))[None, :])[0]
        return [Sphere(s[:3], s[3]) for s in out]

    def filter_self_from_pointcloud(self, pc, point_radius, configuration, environment: Environment | None = None):
        """<robot>.filter_self_from_pointcloud — robot_helper.hh:284-322: the points [m][3] of `pc` whose sphere of radius
        `point_radius` touches neither the robot at `configuration` nor the environment (order preserved)."""
        pts = _f32(pc).reshape(-1, 3)
        q = _f32(configuration, (self._dim,))
        out = np.zeros((max(len(pts), 1), 3), np.float32)
        m = ctypes.c_size_t(0)
        check(lib.vmv_filter_self_from_pointcloud(self._id, self._env(environment), _fp(q), _fp(pts), len(pts),
                                                  float(point_radius), _fp(out), len(pts), ctypes.byref(m)),
              "vmv_filter_self_from_pointcloud")
        return out[:m.value].copy()

    # batched calls --------------------------------------------------------------------------------------------
    def _env(self, environment):
        # the default argument of the reference (`env = Environment()`): one cached empty environment, kept alive
        # here because the C handle must outlive the call
        if environment is None:
            environment = _EMPTY_ENVIRONMENT
        return environment.handle()

    def validate_batch(self, configurations, environment: Environment | None = None):
        """bool[n] (numpy in -> numpy out; torch CUDA tensor in -> torch.bool CUDA tensor out)."""
        if _is_torch_cuda(configurations):
            return self._torch_bits(configurations, None, environment)
        q = _f32(configurations)
        if q.ndim != 2 or q.shape[1] != self._dim:
            raise TypeError(f"expected [n][{self._dim}] configurations")
        n = q.shape[0]
        bits = np.zeros((n + 63) // 64, np.uint64)
        check(lib.vmv_validate_batch_host(self._id, self._env(environment), _fp(q), n,
                                          bits.ctypes.data_as(_lib.c_u64_p)), "vmv_validate_batch_host")
        return unpack_bits(bits, n)

    def validate_motion_batch(self, starts, goals, environment: Environment | None = None):
        if _is_torch_cuda(starts):
            return self._torch_bits(starts, goals, environment)
        a, b = _f32(starts), _f32(goals)
        if a.ndim != 2 or a.shape[1] != self._dim or a.shape != b.shape:
            raise TypeError(f"expected two [n][{self._dim}] arrays")
        n = a.shape[0]
        bits = np.zeros((n + 63) // 64, np.uint64)
        check(lib.vmv_validate_motion_batch_host(self._id, self._env(environment), _fp(a), _fp(b), n,
                                                 bits.ctypes.data_as(_lib.c_u64_p)), "vmv_validate_motion_batch_host")
        return unpack_bits(bits, n)

    def fk_batch(self, configurations):
        """float32 [n][n_spheres][4] = x y z r."""
        q = _f32(configurations)
        if q.ndim != 2 or q.shape[1] != self._dim:
            raise TypeError(f"expected [n][{self._dim}] configurations")
        out = np.zeros((q.shape[0], self._ns, 4), np.float32)
        check(lib.vmv_fk_batch_host(self._id, _fp(q), q.shape[0], _fp(out)), "vmv_fk_batch_host")
        return out

    def halton_device(self, n: int, skip: int = 0):
        """Samples skip+1 .. skip+n of the reference's Halton sequence (random/halton.hh), generated on the GPU.
        Returns a torch float32 CUDA tensor [n][dimension] (bit-exact against the reference's sequence)."""
        import torch

        q = torch.empty((n, self._dim), dtype=torch.float32, device="cuda")
        stream = ctypes.c_void_p(torch.cuda.current_stream(q.device).cuda_stream)
        check(lib.vmv_halton_configs(self._id, int(skip), ctypes.c_void_p(q.data_ptr()), n, stream),
              "vmv_halton_configs")
        return q

    # device-resident (torch tensors are only the memory/stream plumbing) ------------------------------------
    def validate_bits_device(self, q, environment, bits, goals=None):
        """q (and goals): torch float32 CUDA [n][dim] contiguous; bits: torch int64 CUDA [ceil(n/64)].
        Launches on torch's current stream; returns nothing (bits is filled in place)."""
        import torch

        n = q.shape[0]
        assert q.is_contiguous() and q.dtype == torch.float32 and q.shape[1] == self._dim
        assert bits.is_contiguous() and bits.dtype == torch.int64 and bits.numel() >= (n + 63) // 64
        stream = ctypes.c_void_p(torch.cuda.current_stream(q.device).cuda_stream)
        env = self._env(environment)
        if goals is None:
            check(lib.vmv_validate_batch(self._id, env, ctypes.c_void_p(q.data_ptr()), n,
                                         ctypes.c_void_p(bits.data_ptr()), stream), "vmv_validate_batch")
        else:
            assert goals.is_contiguous() and goals.shape == q.shape and goals.dtype == torch.float32
            check(lib.vmv_validate_motion_batch(self._id, env, ctypes.c_void_p(q.data_ptr()),
                                                ctypes.c_void_p(goals.data_ptr()), n, ctypes.c_void_p(bits.data_ptr()),
                                                stream), "vmv_validate_motion_batch")

    def _torch_bits(self, a, b, environment):
        import torch

        with torch.cuda.device(a.device):
            a = a.contiguous().float()
            n = a.shape[0]
            bits = torch.zeros((n + 63) // 64, dtype=torch.int64, device=a.device)
            self.validate_bits_device(a, environment, bits, None if b is None else b.contiguous().float())
            shifts = torch.arange(64, device=a.device, dtype=torch.int64)
            return (((bits[:, None] >> shifts[None, :]) & 1) != 0).reshape(-1)[:n]


_EMPTY_ENVIRONMENT = Environment()

from . import api as _api  # noqa: E402  (the reference's planner-facing names; needs the classes above)
from .api import (AORRTCSettings, FCITNeighborParams, FCITSettings, PRMNeighborParams, PRMSettings,  # noqa: E402,F401
                  RRTCSettings, SimplifyRoutine, SimplifySettings, results_to_dict, DEFAULT_ITERATIONS, ROBOT_RRT_RANGES)

_ROBOT_MODULES = {}
for _name in robots:
    _mod = _Robot(_name)
    _api.install(_mod)
    _ROBOT_MODULES[_name] = _mod
    globals()[_name] = _mod
    sys.modules[_mod.__name__] = _mod
    __all__.append(_name)
__all__ += ["configure_robot_and_planner_with_kwargs", "problem_dict_to_vamp", "results_to_dict", "RRTCSettings",
            "PRMSettings", "PRMNeighborParams", "FCITSettings", "FCITNeighborParams", "AORRTCSettings", "SimplifySettings",
            "SimplifyRoutine"]


def configure_robot_and_planner_with_kwargs(robot_name: str, planner_name: str, **kwargs):
    """vamp.configure_robot_and_planner_with_kwargs (reference src/vamp/__init__.py:69-139)"""
    if robot_name not in _ROBOT_MODULES:
        raise AttributeError(robot_name)
    return _api.configure_robot_and_planner_with_kwargs(_ROBOT_MODULES, robot_name, planner_name, **kwargs)


def problem_dict_to_vamp(problem, ignore_names=()):
    """vamp.problem_dict_to_vamp (reference src/vamp/__init__.py:140-186): MotionBenchMaker scene dict -> Environment.
    Cylinders become capsules, except in the "box" problem where they are over-approximated by cuboids."""
    env = Environment()
    for obj in problem["sphere"]:
        if obj["name"] not in ignore_names:
            s = Sphere(obj["position"], obj["radius"])
            s.name = obj["name"]
            env.add_sphere(s)
    for obj in problem["cylinder"]:
        if obj["name"] in ignore_names:
            continue
        if problem["problem"] == "box":
            c = Cuboid(obj["position"], obj["orientation_euler_xyz"], [obj["radius"], obj["radius"], obj["length"] / 2])
            c.name = obj["name"]
            env.add_cuboid(c)
        else:
            c = Cylinder(obj["position"], obj["orientation_euler_xyz"], obj["radius"], obj["length"])
            c.name = obj["name"]
            env.add_capsule(c)
    for obj in problem["box"]:
        if obj["name"] not in ignore_names:
            c = Cuboid(obj["position"], obj["orientation_euler_xyz"], obj["half_extents"])
            c.name = obj["name"]
            env.add_cuboid(c)
    return env
