"""CPU-only sanitizer job for the host C++ of the product: AddressSanitizer + UBSan and
ThreadSanitizer builds of csrc/{symbolic,ordering,device,gmrfx_api}.cpp, driven through the C ABI by
tools/sanitize_host.cpp (threaded nested dissection, threaded scatter map, concurrent handles, malformed input).
No GPU is involved (GPU sanitizers are not available on the pool)."""
import os
import subprocess

import pytest

PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gaussianmarkovrandomfields.jl_amd")


@pytest.fixture(scope="module")
def built():
    subprocess.check_call(["make", "-s", "-j2", "-C", PKG, "sanitize"])


@pytest.mark.parametrize("which", ["asan", "tsan"])
def test_host_code_is_sanitizer_clean(built, which):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", TSAN_OPTIONS="halt_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([os.path.join(PKG, "build", f"sanitize_{which}")], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "sanitize_host: ok" in r.stdout, (r.stdout[-2000:] + r.stderr[-4000:])
