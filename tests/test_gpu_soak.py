"""A bounded slice of the parity soak (tests/soak_parity.py) inside the driver-run suite: a few hundred seeded random frames,
four frames per handle so that the compared ones are planned with scheduling feedback (strips), a quarter of them also
rendered as 2..8 band or tile shards and stitched.  EXACT precision: visibility bit-exact, RGBA equal to the oracle; FAST
(default) precision: within 1 LSB.  Fixed seeds (a failure reproduces from the commit alone); VF_SOAK_FIRST_SEED moves the window,
and tools/profile_round.sh's long soak runs on fresh seeds every time."""
import os

import pytest

pytestmark = pytest.mark.gpu


def test_soak_slice(oracle):
    import soak_parity
    first = int(os.environ.get("VF_SOAK_FIRST_SEED", "204000"))
    res = soak_parity.run(first=first, cases=400, budget=60.0, verbose=False)
    print("\n" + res["summary"])
    assert res["cases"] >= 40, res["summary"]
    assert not res["bad"], res["bad"][:10]
    assert res["worst_exact"] == 0 and res["worst_fast"] <= 1
    assert res["hist"][2] == 0 and res["hist"][3] == 0
