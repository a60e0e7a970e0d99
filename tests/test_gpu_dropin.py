"""The C launchRays() drop-in inside the real reference program (GPU).

oracle/_ref/dropin_check is the reference's mainTracing.c (compiled in place, its own launchRays renamed)
linked with raytracing_amd/csrc/launch_rays_dropin.c and libgrafthip.so.  It drives the program with the
reference's own scene set-up and key handler and compares, per state, the CPU reference frame with the GPU
drop-in frame byte for byte, and the pixSetRGB call sequences.  The binary is built where the reference
sources exist and travels to the GPU box prebuilt.
"""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BINARY = os.path.join(ROOT, "oracle", "_ref", "dropin_check")


@pytest.mark.gpu
def test_dropin_launchrays_inside_reference_program():
    # (no skip: this is the only test of the drop-in through the reference program itself, and the binary travels to
    # the GPU box with the tree -- `make -C oracle` builds it wherever the reference sources are)
    assert os.path.exists(BINARY), "oracle/_ref/dropin_check is missing: build it with `make -C oracle` where /root/reference exists"
    p = subprocess.run([BINARY], capture_output=True, text=True, timeout=300)
    print(p.stdout)
    print(p.stderr)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "all states identical" in p.stdout and "DIFFERENT" not in p.stdout
