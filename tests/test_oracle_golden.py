"""The CPU oracle against the golden vectors made from the real reference (oracle/gen_golden.py).

Bit-exact: every double of every displayed pixel (FNV-1a64 of the frame + stored pixels), every primary
closest-shape index, the count and order of pixSetRGB calls.  This is what pins the oracle.
"""
import pytest

import oracle_binding as ob
from cases import CASES, build
from golden_util import GOLDEN, check_against_golden


def test_every_case_has_a_golden_vector():
    assert set(CASES) == set(GOLDEN)


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_matches_reference(name):
    st = build(name)
    out = ob.render(st, threads=8, want_index=True)
    check_against_golden(name, st, out["rgb"], out["index"], out["cols"], out["rows"])


def test_survey_probe_values():
    """Numbers the survey of the reference measured independently on the start-up scene (AA off)."""
    g = GOLDEN["default_aa_off"]
    assert g["probe"]["256,256"][0] == "0.5490340272890842"
    assert g["probe"]["40,100"][0] == "0.11794635508752097"
    assert g["idx_hist"] == {"0": 16497, "1": 29034, "2": 3326, "3": 1786, "4": 38269, "7": 173232}
    assert abs(g["max"] - 1.8) < 1e-12 and abs(g["min"] - 0.0156205) < 1e-6
    assert GOLDEN["single_sphere"]["idx_hist"] == {"-1": 245647, "0": 16497}
    assert GOLDEN["cfg3_512"]["idx_hist"]["-1"] == 186006
    assert GOLDEN["cfg4_512"]["idx_hist"]["-1"] == 131772
    assert GOLDEN["cfg5_512"]["idx_hist"]["-1"] == 92738
