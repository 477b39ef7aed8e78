"""The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (CPU only).

oracle/selftest.c drives every oracle entry point once on a small ring -- the reference's own test shapes
(tests/TestBatchedFHEPIE.cpp:89-139, tests/TestOpenFHE.cpp:36-65) at reduced size -- and this test builds it with
-fsanitize=address,undefined and requires a clean exit.  A sanitizer report aborts the program (non-zero exit)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("gcc") is None, reason="no C compiler")
def test_oracle_selftest_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "selftest_san")
    src = [os.path.join(ROOT, "oracle", f) for f in ("selftest.c", "pie_oracle.c", "pie_hashing.c")]
    subprocess.check_call(["gcc", "-O1", "-g", "-std=c11", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-fno-omit-frame-pointer", "-o", exe] + src)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("LD_PRELOAD", None)
    r = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "selftest ok" in r.stdout
