"""CPU: the host side of the C ABI under AddressSanitizer + UndefinedBehaviorSanitizer (the GPU pool offers no device
sanitizers; the HOST code of the library -- argument validation, closed-form topology tables -- is checked here).
tests/native/abi_host_check.cpp is linked against a host-sanitized build of the library and run as a child process."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
SAN = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-g"]


@pytest.mark.skipif(not os.path.exists(CLANG), reason="ROCm clang++ not found")
def test_host_side_of_the_abi_is_clean_under_asan_and_ubsan(tmp_path):
    from echoglad_amd import build
    lib = build.build(extra_flags=SAN + ["-fno-gpu-sanitize"], variant="asan")          # device code unsanitized (not offered here)
    exe = tmp_path / "abi_host_check"
    cmd = [CLANG, *SAN, "-std=c++17", os.path.join(ROOT, "tests", "native", "abi_host_check.cpp"), "-o", str(exe),
           str(lib), f"-Wl,-rpath,{os.path.dirname(lib)}", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:protect_shadow_gap=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([str(exe)], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "0 failure(s)" in r.stdout and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stdout + r.stderr[-2000:]
