"""ctypes binding of the C-ABI library (include/vamp_mvt_amd.h).

The library is the product: if it is missing or does not export every symbol the header declares, importing
this package fails loudly.  There is no Python/CPU fallback for any compute entry point.
"""
from __future__ import annotations

import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
# VMV_LIBRARY: developer override (ablation builds of the same ABI made by tools/); the product is the in-tree library
LIB_PATH = os.environ.get("VMV_LIBRARY") or os.path.join(_HERE, "libvamp_mvt_amd.so")
HEADER_PATH = os.path.join(_HERE, "..", "include", "vamp_mvt_amd.h")

c_float_p = ctypes.POINTER(ctypes.c_float)
c_u64_p = ctypes.POINTER(ctypes.c_uint64)
c_u32_p = ctypes.POINTER(ctypes.c_uint32)
c_size_p = ctypes.POINTER(ctypes.c_size_t)

VMV_OK = 0
STATUS_NAMES = {0: "VMV_OK", 1: "VMV_ERR_INVALID_ARGUMENT", 2: "VMV_ERR_NO_DEVICE", 3: "VMV_ERR_HIP",
                4: "VMV_ERR_CAPACITY", 5: "VMV_ERR_NOT_FINALIZED", 6: "VMV_ERR_UNKNOWN_ROBOT", 7: "VMV_ERR_FINALIZED"}


class VmvError(RuntimeError):
    def __init__(self, status, where, detail=""):
        self.status = status
        super().__init__(f"{where}: {STATUS_NAMES.get(status, status)}" + (f" ({detail})" if detail else ""))


def declared_symbols(header_path: str = HEADER_PATH):
    """Every function name include/vamp_mvt_amd.h declares."""
    with open(header_path) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(vmv_[a-z0-9_]+)\s*\(", text)))


def _load():
    # PyTorch-ROCm bundles its own HIP runtime.  Two HIP runtimes in one process fight over the device
    # ("No HIP GPUs are available" in whichever initialises second), so when torch is installed its runtime is
    # loaded first and this library binds to the same libamdhip64 (torch stays plumbing: memory, streams, RCCL).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python __graft_entry__.py` (hipcc --offload-arch=gfx950). "
            "vamp_mvt_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    if missing:
        raise ImportError(f"{LIB_PATH} does not export: {', '.join(missing)}")
    V, I, F, S = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t
    sig = {
        "vmv_status_string": (ctypes.c_char_p, [I]),
        "vmv_abi_version": (I, []),
        "vmv_last_error": (ctypes.c_char_p, []),
        "vmv_device_count": (I, [ctypes.POINTER(I)]),
        "vmv_set_device": (I, [I]),
        "vmv_get_device": (I, [ctypes.POINTER(I)]),
        "vmv_num_robots": (I, []),
        "vmv_robot_name": (ctypes.c_char_p, [I]),
        "vmv_robot_id": (I, [ctypes.c_char_p]),
        "vmv_robot_dimension": (I, [I]),
        "vmv_robot_n_spheres": (I, [I]),
        "vmv_robot_resolution": (I, [I]),
        "vmv_robot_min_max_radii": (I, [I, c_float_p, c_float_p]),
        "vmv_robot_bounds": (I, [I, c_float_p, c_float_p, c_float_p]),
        "vmv_robot_joint_name": (ctypes.c_char_p, [I, I]),
        "vmv_robot_end_effector": (ctypes.c_char_p, [I]),
        "vmv_env_create": (I, [ctypes.POINTER(V)]),
        "vmv_env_destroy": (I, [V]),
        "vmv_env_add_sphere": (I, [V, F, F, F, F]),
        "vmv_env_add_cuboid": (I, [V, c_float_p]),
        "vmv_env_add_capsule": (I, [V, c_float_p]),
        "vmv_filter_pointcloud": (I, [c_float_p, S, F, F, F, c_float_p, c_float_p, c_float_p, I, I, c_float_p, S, c_size_p,
                                      c_u64_p, c_u64_p]),
        "vmv_env_add_capt_pointcloud_gpu": (I, [V, c_float_p, S, F, F, F, c_u64_p, c_u64_p]),
        "vmv_env_add_heightfield": (I, [V, c_float_p, c_float_p, S, S, c_float_p]),
        "vmv_env_heightfield_count": (I, [V, c_size_p]),
        "vmv_env_add_capt_pointcloud": (I, [V, c_float_p, S, F, F, F, c_u64_p]),
        "vmv_env_add_mvt_pointcloud": (I, [V, c_float_p, S, F, F, c_float_p, c_float_p, F, c_u64_p, ctypes.POINTER(I)]),
        "vmv_env_mvt_count": (I, [V, c_size_p]),
        "vmv_env_mvt_info": (I, [V, S, c_u32_p, c_u32_p, c_u32_p, c_float_p, c_float_p]),
        "vmv_env_finalize": (I, [V]),
        "vmv_env_counts": (I, [V, c_size_p]),
        "vmv_env_get_spheres": (I, [V, c_float_p, S, c_size_p]),
        "vmv_env_get_cuboids": (I, [V, I, c_float_p, S, c_size_p]),
        "vmv_env_get_capsules": (I, [V, I, c_float_p, S, c_size_p]),
        "vmv_env_capt_sizes": (I, [V, S, c_u32_p, c_u32_p]),
        "vmv_env_capt_arrays": (I, [V, S, c_float_p, c_u32_p, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p]),
        "vmv_fk_batch": (I, [I, V, S, V, V]),
        "vmv_validate_batch": (I, [I, V, V, S, V, V]),
        "vmv_validate_batch_env": (I, [I, V, V, S, V, V]),
        "vmv_validate_batch_self": (I, [I, V, S, V, V]),
        "vmv_validate_motion_batch": (I, [I, V, V, V, S, V, V]),
        "vmv_fk_batch_host": (I, [I, c_float_p, S, c_float_p]),
        "vmv_eefk_batch": (I, [I, V, S, V, V]),
        "vmv_contacts_batch_host": (I, [I, V, c_float_p, S, c_u32_p, c_u32_p]),
        "vmv_env_report_layout": (I, [V, c_u32_p]),
        "vmv_robot_self_pairs": (I, [I, c_size_p, ctypes.POINTER(ctypes.c_uint16)]),
        "vmv_eefk_batch_host": (I, [I, c_float_p, S, c_float_p]),
        "vmv_env_attach": (I, [V, c_float_p, c_float_p, S]),
        "vmv_env_detach": (I, [V]),
        "vmv_filter_self_from_pointcloud": (I, [I, V, c_float_p, c_float_p, S, F, c_float_p, S, c_size_p]),
        "vmv_shard_range": (I, [S, I, I, c_size_p, c_size_p]),
        "vmv_shard_words": (S, [S, I]),
        "vmv_spheres_in_collision_batch": (I, [V, V, S, V, V]),
        "vmv_spheres_in_collision_batch_host": (I, [V, c_float_p, S, ctypes.POINTER(ctypes.c_uint8)]),
        "vmv_validate_batch_host": (I, [I, V, c_float_p, S, c_u64_p]),
        "vmv_validate_motion_batch_host": (I, [I, V, c_float_p, c_float_p, S, c_u64_p]),
        "vmv_release_staging": (I, []),
        "vmv_halton_configs": (I, [I, ctypes.c_uint64, V, S, V]),
        "vmv_time_validate_batch": (I, [I, V, V, S, V, I, V, c_float_p]),
        "vmv_fill_uniform_configs": (I, [I, V, S, ctypes.c_uint64, V]),
        "vmv_kernel_name": (ctypes.c_char_p, [I, ctypes.c_char_p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def check(status: int, where: str):
    if status != VMV_OK:
        raise VmvError(status, where, lib.vmv_last_error().decode(errors="replace"))
