"""The index algebra of the long transforms' tile culling, checked on the CPU: tests/host/index_check.cpp includes the product's own
headers (rows_smax_index, plan_long_cols, plan_cols_per_wg are __host__ __device__), is compiled with hipcc for the host and run
here — no GPU, no HIP call. What the GPU tests can only show indirectly (culled == unculled) is pinned directly: the place the rows
kernel writes a run's maximum is the place the plan reads it from, for every run of 65536- and 2^20-point rows; and for the radix-8 /
radix-16 fold's rows (round 5): the layout is a bijection, the transform's epilogue stores every output where the layout says, a
detect tile reads its block in contiguous runs."""
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


def test_the_averager_ring_never_hands_out_rows_that_are_still_to_be_read(tmp_path):
    """csrc/ring_place.h — where a batch's rows go in the averager ring's buffer — against its invariant: 1.2 million calls of random
    streams (tests/host/ring_check.cpp; plain C++, no HIP)."""
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("g++ not found")
    exe = tmp_path / "ring_check"
    subprocess.run([gxx, "-std=c++17", "-O1", "-Wall", "-Werror", "-o", str(exe), os.path.join(ROOT, "tests", "host", "ring_check.cpp")], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.strip().endswith("bad 0")
