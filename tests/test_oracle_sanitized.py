"""CPU sanitizer runs (GPU AddressSanitizer is not available on the pool): the oracle's golden-vector tests against
liboracle_asan.so (-fsanitize=address,undefined), and the C++ host mirror's host-only checks -- builder validation
(csrc/host_logic.hpp behind ndi_validate*), the generic integer strategies -- compiled with the same sanitizers."""
import os
import subprocess
import sys

from conftest import ROOT


def _san_lib(name):
    return subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True, check=True).stdout.strip()


def test_oracle_golden_vectors_under_asan_ubsan():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle_asan.so"], check=True, capture_output=True)
    env = dict(os.environ, LD_PRELOAD=f"{_san_lib('libasan.so')} {_san_lib('libubsan.so')}",
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1",
               ORACLE_LIB=os.path.join(ROOT, "oracle", "liboracle_asan.so"))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle_golden.py"), "-x", "-q",
                        "-p", "no:cacheprovider"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "passed" in r.stdout and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr


def test_cpp_mirror_host_logic_under_asan_ubsan(pkg, tmp_path):
    exe = str(tmp_path / "test_host_mirror_asan")
    libdir = os.path.join(ROOT, "ndarray-interp_amd")
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-o", exe,
                    os.path.join(ROOT, "tests", "cpp", "test_host_mirror.cpp"), "-L", libdir, "-lndinterp_hip",
                    "-Wl,-rpath," + libdir, "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"],
                   check=True, capture_output=True)
    r = subprocess.run([exe, "--host-only"], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1"))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
