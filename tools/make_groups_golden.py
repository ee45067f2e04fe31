#!/usr/bin/env python3
"""Reference-derived collision structure -> tests/golden/groups_<robot>.json (DATA ONLY: small integer tables).

Reads the reference's generated `Robot::fkcc` bodies as data (tools/ref_fk_eval.py: the `// <link>` / `// a vs. b`
comments and the sphere indices of every `sphere_environment_in_collision` / `sphere_sphere_self_collision` call) and
writes, per robot: the environment groups (link, bounding sphere, fine spheres, in visiting order), the self-collision
groups (links a, b, their bounding spheres, the fine pairs in order), the attachment link list of `fkcc_attach`, and
the constants the robot header states (dimension, n_spheres, resolution, radii bounds, joint names).

Oracle and HIP kernels are both generated from vamp_mvt_amd/robots/*.json, so a wrong pair table would be invisible to
HIP-vs-oracle tests; tests/test_robot_structure.py asserts the models against these tables on every CPU run.

Usage (in the container that has /root/reference):  python tools/make_groups_golden.py"""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_fk_eval as R  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")

for name in ("panda", "ur5", "fetch", "baxter"):
    prog = R.load_program(name)
    consts = R.robot_constants(name)
    txt = open(R.ROBOT_HH.format(name=name)).read()
    seg = txt[txt.index("inline static bool fkcc_attach("):]
    seg = seg[seg.index("// attaching at"):]
    seg = seg[:seg.index("return true;")]
    data = {
        "source": f"src/impl/vamp/robots/{name}.hh (Robot::fkcc / fkcc_attach call structure, read as data)",
        "dimension": consts["dimension"], "n_spheres": consts["n_spheres"], "resolution": consts["resolution"],
        "min_radius": consts["min_radius"], "max_radius": consts["max_radius"], "joint_names": consts["joint_names"],
        "n_total_spheres": prog.n_y // 4,
        "env_groups": [[g[0], g[1] // 4, [i // 4 for i in g[2]]] for g in prog.env_groups],
        "self_groups": [[g[0], g[1], g[2][0] // 4, g[2][1] // 4, [[a // 4, b // 4] for a, b in g[3]]]
                        for g in prog.self_groups],
        "attach_frame": re.match(r"// attaching at (\S+)", seg).group(1),
        "attach_links": re.findall(r"// Attachment vs\. (\S+)", seg),
    }
    path = os.path.join(OUT, f"groups_{name}.json")
    with open(path, "w") as f:
        json.dump(data, f, separators=(",", ":"))
    print(name, len(data["env_groups"]), "env groups,", len(data["self_groups"]), "self groups,",
          sum(len(g[4]) for g in data["self_groups"]), "fine pairs,", os.path.getsize(path) // 1024, "KiB")
