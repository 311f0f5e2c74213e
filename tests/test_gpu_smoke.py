"""`__graft_entry__.smoke()` the way the driver runs it at round end: a fresh process WITHOUT the variables tests/conftest.py
pins (tile rect version, rounds) — the renderer's own per-scene choices, and a checker that has to follow them (round 5: the
oracle binding defaults to rect version 4, the renderer picks version 3 for a scene this small, and nothing but the driver
ran smoke())."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_smoke_in_a_clean_environment():
    env = {k: v for k, v in os.environ.items() if not k.startswith("GS3D_")}
    res = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], cwd=ROOT, env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert res.returncode == 0 and "smoke OK" in res.stdout, res.stdout[-3000:]
