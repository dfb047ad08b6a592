"""The C ABI used from plain C++ (HIP runtime only, no Python / torch in the process): examples/cabi_demo packs latent
weights, multiplies, and checks the result bit for bit against its own scalar host loop.  Needs an MI355X."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_demo_runs_without_python_in_the_loop():
    exe = os.path.join(ROOT, "examples", "cabi_demo")
    if not os.path.exists(exe):   # normally built by __graft_entry__.build(); hipcc is in the image
        subprocess.run(["make", "-C", os.path.join(ROOT, "examples")], check=True, timeout=300)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "cabi_demo ok" in r.stdout and r.stdout.count("bit-exact") == 4 and "invalid call rejected" in r.stdout
