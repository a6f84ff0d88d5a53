"""Host side of libgmhip under AddressSanitizer + UBSan + LeakSanitizer: tests/asan/Makefile
compiles the translation units without device code for the host only, links them against a test double of the
HIP runtime and runs tests/asan/asan_driver.cpp -- random valid and invalid programs through validation,
lowering, the program cache and batched launches, the specialiser's source generator, the caching allocator
from 8 threads.  GPU-side sanitizers are not available on the pool."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(900)
def test_host_side_is_clean_under_asan_ubsan():
    if not (os.path.exists("/opt/rocm/bin/hipcc") and shutil.which("make")):
        pytest.skip("hipcc / make not available")
    res = subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "asan"), "run"], capture_output=True, text=True, timeout=850)
    tail = (res.stdout + res.stderr)[-3000:]
    assert res.returncode == 0, tail
    assert "ASAN_DRIVER ok" in res.stdout and "in_use_after=0" in res.stdout, tail
    assert "ERROR: AddressSanitizer" not in tail and "runtime error" not in tail and "LeakSanitizer" not in tail
