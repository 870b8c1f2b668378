"""The C ABI driven from C: tests/native/abi_smoke.c is compiled against include/lrx.h + liblrx.so (no Python or torch in the
process) and must find the exact top-k on both search entry points."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_program_links_and_runs_against_the_abi(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    libdir = os.path.join(ROOT, "lightretriever_amd")
    assert os.path.exists(os.path.join(libdir, "liblrx.so")), "build liblrx.so first (__graft_entry__.build())"
    exe = str(tmp_path / "abi_smoke")
    subprocess.check_call([hipcc, "-x", "hip", "--offload-arch=gfx950", os.path.join(ROOT, "tests", "native", "abi_smoke.c"), "-I" + os.path.join(ROOT, "include"),
                           "-L" + libdir, "-llrx", "-Wl,-rpath," + libdir, "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "ABI SMOKE OK" in out.stdout, out.stdout + out.stderr
