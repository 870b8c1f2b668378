"""A short slice of tools/fuzz_gpu.py in the suite: random GEMM / attention / search shapes and random small encoder
architectures (folded and unfolded norm, bias, both RoPE kinds, MRL) against torch fp32 and the oracle."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [11, 12])
def test_random_shapes(seed):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_gpu.py"), "--seed", str(seed), "--rounds", "18"], capture_output=True, text=True,
                         timeout=600, cwd=ROOT)
    assert out.returncode == 0 and "failures 0" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]
