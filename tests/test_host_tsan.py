"""Thread-sanitizer run of the C-ABI's host layer (helper threads, host-paced stream ordering, present / regrow /
failure protocols) on a fake HIP runtime: tests/host/tsan_host_test.cpp.  CPU only: no GPU, no pixels — order and
error protocol are checked, and ThreadSanitizer must stay silent."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_layer_is_race_free_under_thread_sanitizer(tmp_path):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = tmp_path / "tsan_host_test"
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-I" + os.path.join(ROOT, "tests", "host", "hip_stub"),
           "-x", "c++", os.path.join(ROOT, "software-renderer_amd", "csrc", "swr_api.hip"),
           os.path.join(ROOT, "tests", "host", "hip_stub", "stub_runtime.cpp"),
           os.path.join(ROOT, "tests", "host", "hip_stub", "stub_launch.cpp"),
           os.path.join(ROOT, "tests", "host", "tsan_host_test.cpp"), "-lpthread", "-o", str(exe)]
    build = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    if build.returncode != 0 and "tsan" in build.stderr.lower() and "cannot find" in build.stderr.lower():
        pytest.skip("libtsan is not installed")
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66"))
    out = run.stdout + run.stderr
    assert "WARNING: ThreadSanitizer" not in out, out[-4000:]
    assert run.returncode == 0 and "tsan host test: ok" in out, out[-2000:]
