#!/usr/bin/env python3
"""MotionBenchMaker problem archives of the reference -> tests/golden/mbm_<robot>.npz (DATA ONLY).

The reference holds one fixture family that exercises capsules and rotated cuboids: resources/<robot>/problems.tar.bz2
(MoveIt scene + request YAML per problem), with published validity counts (resources/README.md:146,81,210):
699 / 700 Panda, 608 / 700 UR5, 679 / 700 Fetch problems have a valid start and a valid goal.

This script reads the YAML (yaml.safe_load; nothing in the archive is executed), applies the scene -> primitive mapping of
resources/problem_tar_to_pkl_json.py:33-77 (pose = collision-object pose x primitive pose; position; static-xyz Euler
angles; box half extents = dimensions / 2; cylinder length, radius) and writes, per robot, flat arrays:

    names[n]            problem family of problem i ("box" selects the cylinder -> cuboid rule of src/vamp/__init__.py:153-165)
    index[n]            the problem's number inside its family
    start[n][dim], goal[n][dim]     joint values in the robot's joint order
    sphere_off[n+1], spheres[*][4]      x y z r
    cyl_off[n+1],    cylinders[*][8]    x y z | euler xyz | radius | length
    box_off[n+1],    boxes[*][9]        x y z | euler xyz | half extents

The Euler angles are recomputed here from the quaternions (standard formulas, double precision), so the pin is
tolerance-level: the primitives handed to the kernels equal the reference's up to the rounding of that conversion.

Usage (in the container that has /root/reference):  python tools/make_mbm_golden.py"""
import math
import os
import re
import sys
import tarfile

import numpy as np
import yaml

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
REF = os.environ.get("VAMP_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)


def pose_matrix(tf):
    """4 x 4 from {'position': xyz, 'orientation': quaternion x y z w}"""
    x, y, z, w = (float(v) for v in tf["orientation"])
    n = x * x + y * y + z * z + w * w
    m = np.identity(4)
    if n > 1e-30:
        s = 2.0 / n
        m[:3, :3] = [[1 - s * (y * y + z * z), s * (x * y - z * w), s * (x * z + y * w)],
                     [s * (x * y + z * w), 1 - s * (x * x + z * z), s * (y * z - x * w)],
                     [s * (x * z - y * w), s * (y * z + x * w), 1 - s * (x * x + y * y)]]
    m[:3, 3] = [float(v) for v in tf["position"]]
    return m


def euler_xyz(m):
    """static-frame x, y, z angles: R = Rz(c) Ry(b) Rx(a)"""
    cy = math.hypot(m[0, 0], m[1, 0])
    if cy > 4 * np.finfo(float).eps:
        return math.atan2(m[2, 1], m[2, 2]), math.atan2(-m[2, 0], cy), math.atan2(m[1, 0], m[0, 0])
    return math.atan2(-m[1, 2], m[1, 1]), math.atan2(-m[2, 0], cy), 0.0


def scene_objects(data):
    out = {"sphere": [], "cylinder": [], "box": []}
    for co in data["world"]["collision_objects"]:
        base = pose_matrix(co["pose"]) if "pose" in co else np.identity(4)
        prim = co["primitives"][0]
        m = base @ pose_matrix(co["primitive_poses"][0])
        pos, eul, dims = m[:3, 3].tolist(), list(euler_xyz(m)), [float(d) for d in prim["dimensions"]]
        if prim["type"] == "sphere":
            out["sphere"].append(pos + [dims[0]])
        elif prim["type"] == "cylinder":
            out["cylinder"].append(pos + eul + [dims[1], dims[0]])  # radius, length
        elif prim["type"] == "box":
            out["box"].append(pos + eul + [d / 2 for d in dims])
        else:
            raise RuntimeError(prim["type"])
    return out


def request_configs(data, joints):
    js = data["start_state"]["joint_state"]
    start = [float(js["position"][js["name"].index(j)]) for j in joints]
    goal_c = {e["joint_name"]: float(e["position"]) for e in data["goal_constraints"][0]["joint_constraints"]}
    return start, [goal_c[j] for j in joints]


def main():
    import vamp_mvt_amd as vamp  # joint order only (no GPU needed)

    for robot in ("panda", "ur5", "fetch"):
        joints = getattr(vamp, robot).joint_names()
        scenes, requests = {}, {}
        with tarfile.open(os.path.join(REF, "resources", robot, "problems.tar.bz2"), "r:bz2") as tar:
            for member in tar.getmembers():
                if not member.isfile():
                    continue
                _, family, filename = member.name.split("/")
                family = family.replace(f"_{robot}", "")
                key = (family, int(re.findall(r"\d+", filename)[0]))
                data = yaml.safe_load(tar.extractfile(member).read())
                if "scene" in filename:
                    scenes[key] = scene_objects(data)
                elif "request" in filename:
                    requests[key] = request_configs(data, joints)
        keys = sorted(scenes)
        assert keys == sorted(requests), "scene / request files do not pair up"
        arrays = {"names": np.array([k[0] for k in keys]), "index": np.array([k[1] for k in keys], np.int32),
                  "start": np.array([requests[k][0] for k in keys], np.float64),
                  "goal": np.array([requests[k][1] for k in keys], np.float64)}
        for kind, width, off, name in (("sphere", 4, "sphere_off", "spheres"), ("cylinder", 8, "cyl_off", "cylinders"),
                                       ("box", 9, "box_off", "boxes")):
            counts = [len(scenes[k][kind]) for k in keys]
            arrays[off] = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
            rows = [row for k in keys for row in scenes[k][kind]]
            arrays[name] = np.array(rows, np.float64).reshape(-1, width)
        path = os.path.join(ROOT, "tests", "golden", f"mbm_{robot}.npz")
        np.savez_compressed(path, **arrays)
        print(robot, len(keys), "problems;", {k: int(v[-1]) for k, v in arrays.items() if k.endswith("_off")},
              f"{os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
