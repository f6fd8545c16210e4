"""A compiled C++ host (g++, no hipcc, no Python in the data path) drives the hot path through the
C ABI on the GPU — the shape of the Rust/C++ integration INTEGRATION.md describes."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build_and_run(pkg, tmp_path, name):
    pkg.load_library()
    exe = tmp_path / name
    libdir = os.path.dirname(pkg.LIB_PATH)
    subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", name + ".cpp"), "-o", str(exe), "-L", libdir, "-lglprover",
                    "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    r = subprocess.run([str(exe)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout


def test_cpp_host_roundtrip(pkg, tmp_path):
    _build_and_run(pkg, tmp_path, "host_roundtrip")


def test_cpp_host_mapreduce_path(pkg, tmp_path):
    """circuit setup, leaf proofs with public inputs, RCCL exchange, statement-bound verification, verdict all-reduce, digests — through the
    C ABI only, from a compiled host: the MapReduce path a Rust host would run"""
    _build_and_run(pkg, tmp_path, "host_mapreduce")
