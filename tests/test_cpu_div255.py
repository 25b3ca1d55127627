"""The augmentation kernel divides by 255 with a 3-instruction sequence instead of the IEEE division
(drqv2_amd/csrc/elementwise.hip: div255).  tools/check_div255.py restates the sequence in exact arithmetic and checks
it against the correctly rounded quotient for every float32 mantissa; this test runs that check."""
import os
import runpy

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_three_instruction_division_by_255_is_correctly_rounded(capsys):
    runpy.run_path(os.path.join(ROOT, "tools", "check_div255.py"), run_name="__main__")
    assert "mismatches: 0" in capsys.readouterr().out
