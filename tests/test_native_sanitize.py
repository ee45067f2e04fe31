"""The product's host-side builders (CAPT, MVT, broad-phase grid: plain C++ headers) compiled with g++ under
AddressSanitizer + UndefinedBehaviorSanitizer and run on synthetic clouds; the CAPT arrays are compared with the oracle's
in the same program.  CPU only (GPU sanitizers are not available on this pool)."""
import os
import subprocess

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def test_host_builders_under_sanitizers(oracle):  # the fixture makes sure oracle/gen exists
    out = os.path.join(ROOT, "build", "builders_sanitize")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(["gcc", "-c", "-O1", "-g", "-std=c11", "-ffp-contract=off", "-fsanitize=address,undefined",
                           "-fno-omit-frame-pointer", os.path.join(ROOT, "oracle", "vamp_oracle.c"), "-o", out + "_oracle.o"])
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-ffp-contract=off", "-fsanitize=address,undefined",
                           "-fno-omit-frame-pointer", os.path.join(ROOT, "tests", "native", "builders_sanitize.cc"),
                           out + "_oracle.o", "-o", out, "-lm", "-lpthread"])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([out], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout[-2000:] + r.stderr[-4000:]
