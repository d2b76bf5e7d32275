"""Host side of the library under ASan+UBSan and TSan (CPU build; `make asan`, `make tsan`): the staging
copy pool hammered from several threads, the codec-free stream state machines, and the no-device paths of
the C ABI.  GPU sanitizers are not available on this pool; the kernels are covered by the parity tests."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("target", ["asan", "tsan"])
def test_host_library_under_sanitizer(target):
    if shutil.which("g++") is None or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        pytest.skip("g++ or the HIP host headers are not available")
    r = subprocess.run(["make", "-C", ROOT, target], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert "host_san_test ok" in r.stdout
    for bad in ("ERROR: AddressSanitizer", "WARNING: ThreadSanitizer", "runtime error:"):
        assert bad not in r.stdout + r.stderr
