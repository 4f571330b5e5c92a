"""Build every native artefact of amplisolve_amd in-tree.

  lib/libamplisolve_hip.so   HIP kernels + C ABI (include/amplisolve_hip.h), hipcc --offload-arch=gfx950
  lib/libamplisolve_host.so  C++ host: BED / ASEQ / error-table parsers, SoA packer, writers (include/amplisolve_host.h)
  bin/AmpliSolveErrorEstimation, bin/AmpliSolveVariantCalling   the two drop-in command lines

hipcc cross-compiles gfx950 without a GPU, so this runs in the CPU-only container too.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "lib")
BIN = os.path.join(PKG, "bin")

HIP_LIB = os.path.join(LIB, "libamplisolve_hip.so")
HOST_LIB = os.path.join(LIB, "libamplisolve_host.so")

# ampli_kernels.hip: the path's kernels + their C ABI; ampli_pileup.hip: the upstream counting kernel; ampli_comm.hip: RCCL binding
HIP_SOURCES = ["ampli_kernels.hip", "ampli_pileup.hip", "ampli_comm.hip"]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-shared", "-std=c++17", "-ffp-contract=off"]
CXX_FLAGS = ["-O2", "-fPIC", "-std=c++17", "-ffp-contract=off", "-Wall", "-pthread"]


def _newer(target: str, sources: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources if os.path.exists(s))


def _run(cmd: list[str]) -> None:
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout + r.stderr)
        raise RuntimeError(f"build step failed: {cmd[0]} (exit {r.returncode})")


def hipcc_path() -> str:
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found; the HIP kernels cannot be built")


def build_hip(force: bool = False) -> str:
    os.makedirs(LIB, exist_ok=True)
    srcs = [os.path.join(CSRC, f) for f in HIP_SOURCES]
    deps = srcs + [os.path.join(CSRC, "ampli_internal.h"), os.path.join(CSRC, "ampli_math.h"), os.path.join(CSRC, "ampli_synth.h"),
                   os.path.join(ROOT, "include", "amplisolve_hip.h")]
    if force or _newer(HIP_LIB, deps):
        _run([hipcc_path(), *HIPCC_FLAGS, "-o", HIP_LIB, *srcs])
    return HIP_LIB


def build_host(force: bool = False) -> str:
    os.makedirs(LIB, exist_ok=True)
    os.makedirs(BIN, exist_ok=True)
    hdir = os.path.join(CSRC, "host")
    srcs = [os.path.join(hdir, f) for f in sorted(os.listdir(hdir)) if f.endswith(".cpp") and not f.endswith("_main.cpp")]
    deps = srcs + [os.path.join(hdir, f) for f in os.listdir(hdir) if f.endswith(".hpp")] + [
        os.path.join(CSRC, "ampli_math.h"), os.path.join(CSRC, "ampli_synth.h"),
        os.path.join(ROOT, "include", "amplisolve_host.h"), os.path.join(ROOT, "include", "amplisolve_hip.h")]
    if force or _newer(HOST_LIB, deps):
        _run(["g++", *CXX_FLAGS, "-shared", "-o", HOST_LIB, *srcs, "-ldl", "-lz"])
    for exe, main in (("AmpliSolveErrorEstimation", "ee_main.cpp"), ("AmpliSolveVariantCalling", "vc_main.cpp"), ("computeCounts", "cc_main.cpp")):
        msrc = os.path.join(hdir, main)
        out = os.path.join(BIN, exe)
        if os.path.exists(msrc) and (force or _newer(out, deps + [msrc, HOST_LIB])):
            _run(["g++", *CXX_FLAGS, "-o", out, msrc, "-L" + LIB, "-lamplisolve_host", "-ldl",
                  "-Wl,-rpath,$ORIGIN/../lib"])
    return HOST_LIB


def build_oracle() -> None:
    """Test infrastructure: the CPU oracle and (when /root/reference exists) the reference builds."""
    _run(["make", "-s", "-C", os.path.join(ROOT, "oracle")])


def build_all(force: bool = False) -> None:
    build_hip(force)
    build_host(force)
    build_oracle()


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
    print("built:", HIP_LIB, HOST_LIB)
