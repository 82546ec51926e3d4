"""The committed recipe of the golden vectors reproduces them.

tests/golden/make_golden.py imports the reference (read-only, /root/reference) and regenerates every fixture; `--check`
writes into a scratch directory and compares with the committed files (arrays bit for bit, JSON leaf by leaf, NaN == NaN).
The reference tree exists only in the build container, so the test skips on the GPU box."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isdir("/root/reference/third_party_methods"), reason="needs the reference tree")
def test_default_all_steps_run_regenerates_every_golden_file_identically():
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "make_golden.py"), "--check"],
                       capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "golden check ok: 13 files regenerate identically" in r.stdout
