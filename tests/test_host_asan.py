"""AddressSanitizer + UBSan build of the HOST side of the C-ABI library.  hipcc compiles the
device code as usual and instruments only the host code (-Xarch_host -fsanitize=...); the driver then runs every
entry point's argument validation and the host-only algebra (pyramid operators, LU inverse, compositional update)
without launching a kernel, so it runs without a GPU.  GPU-side sanitizers are not available on this pool."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_code_under_address_and_ub_sanitizers(tmp_path):
    exe = str(tmp_path / "asan_driver")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O1", "-g", "-std=c++17",
           "-Xarch_host", "-fsanitize=address", "-Xarch_host", "-fsanitize=undefined",
           "-Xarch_host", "-fno-omit-frame-pointer", "-Xarch_host", "-fno-sanitize-recover=all", "-o", exe,
           os.path.join(ROOT, "tests", "host", "asan_driver.cpp"),
           os.path.join(ROOT, "inverse_compositional_algorithm_amd", "csrc", "ica_api.hip")]
    subprocess.run(cmd, check=True, capture_output=True, text=True, timeout=900)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "ASAN_DRIVER_OK" in r.stdout
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr
