"""The upload engine (pdb_eda_amd/csrc/pdbeda_upload.h) on a host stand-in for the HIP runtime: tests/upload_harness.cpp.

CPU suite: the same scenarios the GPU suite runs on the device (tests/test_gpu_multiple.py: six threads uploading files of many
sizes at once; a context whose deadline passes in the middle of an upload beside two that keep uploading) plus the one the device
cannot stage -- a copy that never completes (one bad copy must not hold every later upload of the process).
tools/sanitize_cpu.sh runs the three under ThreadSanitizer and AddressSanitizer + UBSan."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("upload") / "upload_harness")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-pthread", "-Wall", "-I" + os.path.join(ROOT, "pdb_eda_amd", "csrc"),
                           os.path.join(ROOT, "tests", "upload_harness.cpp"), "-o", exe])
    return exe


@pytest.mark.parametrize("scenario", ["many", "deadline", "stall"])
def test_upload_engine_on_the_host_stand_in(harness, scenario):
    proc = subprocess.run([harness, scenario], capture_output=True, text=True, timeout=300)
    assert proc.returncode == 0, proc.stdout + proc.stderr
    assert scenario + ":" in proc.stdout
