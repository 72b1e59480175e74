"""The index algebra of the long transforms' tile culling, checked on the CPU: tests/host/index_check.cpp includes the product's own
headers (rows_smax_index, plan_long_cols, plan_cols_per_wg are __host__ __device__), is compiled with hipcc for the host and run
here — no GPU, no HIP call. What the GPU tests can only show indirectly (culled == unculled) is pinned directly: the place the rows
kernel writes a run's maximum is the place the plan reads it from, for every run of 65536- and 2^20-point rows."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_smax_index_and_plan_shapes(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not found")
    exe = tmp_path / "index_check"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O1", "-std=c++17", "-o", str(exe), os.path.join(ROOT, "tests", "host", "index_check.cpp")], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "bad 0" in out.stdout.splitlines()[-1]
