"""The boundary is a C ABI: examples/c_abi_search.cpp is a host program with no Python and no torch in it (ivr_api.h + the
HIP runtime + libivr_hip.so).  Compile it with hipcc and run it: it checks ids and scores against its own f64 brute force."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "intelligent-video-analysis-retrieval-system_amd", "lib")


def test_plain_cpp_host_program(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "c_abi_search")
    build = subprocess.run([hipcc, "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_abi_search.cpp"),
                            "-L" + LIB, "-livr_hip", "-Wl,-rpath," + LIB, "-o", exe], capture_output=True, text=True, timeout=600)
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0 and "0 id mismatches" in run.stdout, run.stdout + run.stderr[-2000:]
