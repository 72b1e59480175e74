"""Build libspecscan.so (HIP, gfx950) in-tree with hipcc. hipcc cross-compiles without a GPU."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libspecscan.so")
# the same sources with -DSS_DIAG: implementation choices can be overridden from the environment at ss_create / sc_create.
# Used by the A/B tests and the measurement scripts only; the product library never reads the environment.
LIB_DIAG = os.path.join(CSRC, "libspecscan_diag.so")
SOURCES = ["specscan.hip", "channelizer.hip"]
HEADERS = ["fft_kernels.h", "fft8192_kernel.h", "fft8192_v2.h", "scan_step.h", "fft256_kernels.h", "detect_kernels.h", "detect_fused.h", "reference_nan.h", "fft1024_kernels.h", "ring_place.h",
           os.path.join("..", "..", "include", "specscan.h"), os.path.join("..", "..", "include", "specscan_channelizer.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-fPIC", "-shared", "-Wall", "-Wno-unused-result"]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP engine cannot be built (there is no CPU fallback)")


def needs_build(lib: str = LIB) -> bool:
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_variant(tag: str, defines, force: bool = False, verbose: bool = False) -> str:
    """A/B builds for the measurement scripts: the diagnostics build with extra -D switches (cache-policy bits of the frame
    loads and so on) -> scripts/ab/libspecscan_<tag>.so (git-ignored; travels to the GPU box). Never loaded by the product."""
    out_dir = os.path.join(HERE, "..", "scripts", "ab")
    os.makedirs(out_dir, exist_ok=True)
    lib = os.path.abspath(os.path.join(out_dir, f"libspecscan_{tag}.so"))
    if force or needs_build(lib):
        cmd = [_hipcc(), *FLAGS, "-DSS_DIAG", *[f"-D{d}" for d in defines], "-o", lib, *[os.path.join(CSRC, s) for s in SOURCES]]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True, cwd=CSRC)
    return lib


def build_lib(force: bool = False, verbose: bool = False, diag: bool = False) -> str:
    lib = LIB_DIAG if diag else LIB
    if force or needs_build(lib):
        cmd = [_hipcc(), *FLAGS, *(["-DSS_DIAG"] if diag else []), "-o", lib, *[os.path.join(CSRC, s) for s in SOURCES]]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True, cwd=CSRC)
    return lib


HOST_DIR = os.path.join(HERE, "host")
HOST_LIB = os.path.join(HOST_DIR, "libspecscan_host.so")


def build_host_lib(force: bool = False, verbose: bool = False) -> str:
    """g++ -> host/libspecscan_host.so: the host-side signal tracker and the raw dump files (no GPU code)."""
    srcs = [os.path.join(HOST_DIR, f) for f in ("signal_tracker.cpp", "raw_file.cpp")]
    deps = srcs + [os.path.join(HOST_DIR, f) for f in ("signal_tracker.h", "raw_file.h")]
    if force or not os.path.exists(HOST_LIB) or max(os.path.getmtime(d) for d in deps) > os.path.getmtime(HOST_LIB):
        cmd = [shutil.which("g++") or "g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-Wall", "-Wextra", "-Wpedantic", "-Werror", "-o", HOST_LIB,
               *srcs, "-lpthread"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True, cwd=HOST_DIR)
    return HOST_LIB


REPLAY_TOOL = os.path.join(HOST_DIR, "specscan_replay")


def build_replay_tool(force: bool = False, verbose: bool = False) -> str:
    """g++ -> host/specscan_replay: the C++ replay host (host/replay_main.cpp) linked against libspecscan.so."""
    src = os.path.join(HOST_DIR, "replay_main.cpp")
    deps = [src, os.path.join(HOST_DIR, "raw_file.h"), os.path.join(HERE, "..", "include", "specscan.h"), LIB]
    if force or not os.path.exists(REPLAY_TOOL) or max(os.path.getmtime(d) for d in deps) > os.path.getmtime(REPLAY_TOOL):
        build_lib()
        cmd = [shutil.which("g++") or "g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-Wpedantic", "-Werror",
               "-I" + os.path.join(HERE, "..", "include"), "-o", REPLAY_TOOL, src, "-L" + CSRC, "-lspecscan", "-Wl,-rpath,$ORIGIN/../csrc",
               "-Wl,-rpath-link," + os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib"), "-lpthread"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True, cwd=HOST_DIR)
    return REPLAY_TOOL


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
    print(build_lib(force=True, verbose=True, diag=True))
    print(build_host_lib(force=True, verbose=True))
    print(build_replay_tool(force=True, verbose=True))
