"""Round 6, VERDICT r05 item 1 step 0: the image-order visibility walk (per pixel, descending primitive id, first exact hit wins) was
modelled on the CPU before any kernel was written (tests/walk_model/).  The model's enumeration of candidate cells must be COMPLETE
-- its winners equal the oracle's visibility -- or its step counts (profiles/r06_walk_step0.log: the no-go) would be meaningless."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "walk_model"))


def test_the_walk_model_finds_the_oracles_winner_at_every_pixel(capsys):
    import walk_step0
    for argv in (["--size", "256x192", "--grid", "160", "--sample", "1", "--seed", "5"],
                 ["--size", "192x192", "--grid", "96", "--sample", "1", "--camera", "fill", "--seed", "6"],
                 ["--size", "256x144", "--grid", "200", "--sample", "1", "--camera", "orbit", "--pose", "37", "--seed", "7"],
                 ["--size", "200x200", "--grid", "64", "--sample", "1", "--smooth"]):
        assert walk_step0.main(argv) == 0, argv
    out = capsys.readouterr().out
    assert out.count("mismatches 0 of") == 4
