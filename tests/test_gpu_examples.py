"""examples/notebook_loops.py stays runnable: three of the reference's notebooks with the imports changed, run on the GPU
through the single-instance API and compared with the figures the notebooks store."""
import os
import re
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_the_notebook_loops_example_retraces_the_stored_figures(capsys):
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import notebook_loops
    notebook_loops.cart_on_track()
    notebook_loops.double_pendulum()
    notebook_loops.ur5_to_a_point()
    out = capsys.readouterr().out
    assert "Distance to UR5Home pos: 1.0192" in out                  # (the text the notebook stores)
    misses = [float(v) for v in re.findall(r"([0-9]+\.[0-9]+) px", out) + re.findall(r"([0-9]+\.[0-9]+) / [0-9.]+ px", out)]
    misses += [float(v) for v in re.findall(r"[xyz] ([0-9]+\.[0-9]+)", out)]
    assert len(misses) >= 9 and max(misses) < 1.0, (misses, out)
