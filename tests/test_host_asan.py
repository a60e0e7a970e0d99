"""The host side of the library (packJob and the tables it builds, the pure-host entry points) under AddressSanitizer
and UndefinedBehaviorSanitizer: tests/host_asan/host_pack_check.cpp is compiled with g++ -- no HIP involved -- and
run over well-formed random scenes at every screen cap and group size and over hostile inputs (negative counts,
avoid slices outside their array, NaN / infinite / huge coordinates, 0 and 1025 passes, one-column frames, 4096
coincident spheres).  The contract is an error code, never a wild access; every packed job is also checked for
offsets, list entries and grid cells that point outside what was packed."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_asan")


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++ with the sanitizer run-times")
def test_host_packers_under_sanitizers():
    subprocess.run(["make", "-C", HERE], check=True, capture_output=True, text=True)
    run = subprocess.run([os.path.join(HERE, "host_pack_check"), "240"], capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-4000:]
    assert "host pack check ok" in run.stdout
