#!/bin/bash
# A/B: the two-kernel path (default) vs the fused one-FK kernel (VMV_FUSED_KERNEL=1), Panda and UR5, 1M configurations
cat > /tmp/ab_fused.py <<'PY'
import ctypes, sys, torch
sys.path.insert(0, ".")
import vamp_mvt_amd as vamp
from vamp_mvt_amd.workloads import environment_from_spec, shell_spec
from vamp_mvt_amd._lib import check, lib
vamp.set_device(0)
env = environment_from_spec(shell_spec(0))
n = 1 << 20
for name in ("panda", "ur5"):
    mod = getattr(vamp, name)
    q = torch.empty((n, mod.dimension()), device="cuda")
    s = torch.cuda.current_stream()
    check(lib.vmv_fill_uniform_configs(mod._id, ctypes.c_void_p(q.data_ptr()), n, 7, ctypes.c_void_p(s.cuda_stream)), "fill")
    bits = torch.zeros(n // 64, dtype=torch.int64, device="cuda")
    for _ in range(20):
        mod.validate_bits_device(q, env, bits)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        mod.validate_bits_device(q, env, bits)
    e1.record()
    torch.cuda.synchronize()
    print(f"  {name}: {e0.elapsed_time(e1) / 100:.4f} ms per 1M configurations, valid {float(vamp.unpack_bits(bits.cpu().numpy().view('uint64'), n).mean()):.4f}")
PY
echo "two kernels (default)"; python /tmp/ab_fused.py 2>/dev/null
echo "fused kernel (VMV_FUSED_KERNEL=1)"; VMV_FUSED_KERNEL=1 python /tmp/ab_fused.py 2>/dev/null
