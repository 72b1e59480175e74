"""The register budget of the step kernel, checked without a GPU: tests/host/step_kernel_resources.hip instantiates the two
instantiations that matter (8192 points CF32 — what bench.py times — and the long transforms' int8 column launch), hipcc compiles
them for gfx950 with the product's code-generation flags and reports what the kernels use. 512 threads at <= 64 VGPRs are eight
waves per SIMD = four workgroups per CU whatever their roles; the spill figures are the ones the measured
numbers were taken with — one careless loop in a role cost eleven more spilled registers and 12 % of the step in round 3."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_step_kernel_keeps_its_registers(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not found")
    import rtl_sdr_scanner_cpp_amd as pkg
    codegen = [f for f in pkg.build.FLAGS if f.startswith(("--offload-arch", "-O", "-std", "-f")) and f not in ("-fPIC",)]
    out = subprocess.run([hipcc, *codegen, "--cuda-device-only", "-c", "-Rpass-analysis=kernel-resource-usage", "-o", str(tmp_path / "k.o"),
                          os.path.join(ROOT, "tests", "host", "step_kernel_resources.hip")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    seen = {}
    for block in re.split(r"remark: [^\n]*Function Name: ", out.stderr)[1:]:
        name = block.split(" ")[0]
        if "k_scan_step" not in name:
            continue
        get = lambda key: int(re.search(key + r": (\d+)", block).group(1))  # noqa: E731
        seen[name] = dict(vgprs=get("VGPRs"), spill=get("VGPRs Spill"), scratch=get(r"ScratchSize \[bytes/lane\]"), occupancy=get(r"Occupancy \[waves/SIMD\]"))
    assert len(seen) == 2, seen
    for name, r in seen.items():
        assert r["vgprs"] <= 64 and r["occupancy"] == 8, (name, r)
        assert r["spill"] <= 11 and r["scratch"] <= 32, (name, r)  # what every number of DESIGN.md 4.1 / 4.4 was measured with
