"""CPU tier: the packet shim's host code (csrc/fpga_shim.cpp: service threads, queues, pinned-buffer pool, index hand-over,
packet assembly, split-and-retry) under ThreadSanitizer and AddressSanitizer.  GPU sanitizers are not available on the pool, so
the shim is built with g++ against test-only stand-ins for the device calls (tests/shim_stub/) and driven the way the reference
drives its driver: six producer threads (map.c:439-444), one receiver (fpga_chaindp.c:228-266), both packet kinds, an index
image replaced in mid-stream, a 1 MiB in-flight budget (the NULL / retry path) and -- second run -- a batch capacity small enough
for splits and err_flag = 1 answers.  The stress driver checks every result packet; a sanitizer report fails the run."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "shim_stub")


@pytest.fixture(scope="module")
def built():
    if not shutil.which("g++"):
        pytest.skip("no g++")
    r = subprocess.run(["make", "-s", "all"], cwd=HERE, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        pytest.skip("sanitizer build not possible here: " + r.stdout[-400:])
    return HERE


@pytest.mark.parametrize("binary", ["shim_tsan", "shim_asan"])
@pytest.mark.parametrize("hits,cap", [(1, 32 << 20), (8, 1500)])
def test_shim_host_code_is_clean_under_sanitizers(built, binary, hits, cap):
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66", ASAN_OPTIONS="detect_leaks=1 exitcode=67", UBSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([os.path.join(built, binary), str(hits), str(cap)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "WARNING: ThreadSanitizer" not in r.stdout and "ERROR: AddressSanitizer" not in r.stdout and "runtime error" not in r.stdout, r.stdout[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("packets ")][-1].split()
    stats = dict(zip(line[0::2], line[1::2])) if False else None
    assert " bad 0 " in r.stdout and "live_indexes 0" in r.stdout
    if cap < 10000:
        assert " err 0 " not in r.stdout            # the small capacity must have produced err_flag answers and splits
