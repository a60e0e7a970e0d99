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


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++ with the sanitizer run-times")
def test_tables_do_not_depend_on_the_number_of_host_threads():
    """The tables of large scenes are built on several host threads (GRAFT_HOST_THREADS caps them): every packed image of
    the same scenes must come out byte for byte the same on one thread and on five."""
    import re
    subprocess.run(["make", "-C", HERE], check=True, capture_output=True, text=True)
    seen = []
    for threads in ("1", "5"):
        run = subprocess.run([os.path.join(HERE, "host_pack_check"), "48"], capture_output=True, text=True, timeout=900,
                             env=dict(os.environ, GRAFT_HOST_THREADS=threads))
        assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-4000:]
        m = re.search(r"images fnv ([0-9a-f]{16}) \((\d+) host threads\)", run.stdout)
        assert m and m.group(2) == threads, run.stdout[-500:]
        seen.append(m.group(1))
    assert seen[0] == seen[1]
