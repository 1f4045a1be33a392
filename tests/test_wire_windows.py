"""The wire layer's request windows (pir_amd/csrc/wire.cpp: two windows in flight per context, flat combining over the
two batch sets, key-set pins and handles, the worker pool) under ThreadSanitizer and AddressSanitizer on the CPU:
wire.cpp + wire_codec.cpp linked against the mock device backend of tests/cpp/wire_windows_test.cpp (an executor thread
that plays the GPU's in-order streams with random delays), driven by one thread and by eight, with more clients than key
set slots.  No GPU, no oracle: the mock's reply is a checksum of (query, client's keys, reply index)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOURCES = [os.path.join(ROOT, "tests", "cpp", "wire_windows_test.cpp"),
           os.path.join(ROOT, "pir_amd", "csrc", "wire.cpp"), os.path.join(ROOT, "pir_amd", "csrc", "wire_codec.cpp")]


@pytest.mark.parametrize("sanitizer", ["thread", "address,undefined"])
def test_request_windows_under_sanitizers(tmp_path, sanitizer):
    exe = str(tmp_path / "wire_windows_test")
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=" + sanitizer, "-fno-omit-frame-pointer"] + SOURCES +
                   ["-o", exe, "-lpthread"], check=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0", TSAN_OPTIONS="halt_on_error=1")   # the pool and the response
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=900)          # buffer cache are kept on purpose
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-6000:]
    assert "wire_windows_test OK" in r.stdout
    assert "ThreadSanitizer" not in r.stderr and "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr
