"""The C-ABI boundary: the shared library loads and exports every symbol include/*.h
declares; without a GPU the product fails loudly instead of falling back to the CPU."""
import ctypes
import glob
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = []
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        src = open(h).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names += re.findall(r"\b(glp_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg.load_library()
    syms = declared_symbols()
    assert len(syms) >= 20
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, f"declared in include/ but not exported: {missing}"
    assert b"gfx950" in lib.glp_version()


def test_no_cpu_fallback(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu tests")
    with pytest.raises(pkg.GlpError):
        pkg.Prover(0)


def test_product_does_not_reference_oracle():
    """nothing under the package may import, link or mention the oracle"""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "0-kno-blobstreamx_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cuh", ".h", ".cpp", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liboracle" not in txt and "orc_" not in txt and "oracle/" not in txt, os.path.join(dirpath, f)


def test_header_is_plain_c_and_links(pkg, tmp_path):
    """include/glprover.h compiles as strict C99 and every symbol used links against the library"""
    import subprocess
    pkg.load_library()
    exe = tmp_path / "abi_c99"
    libdir = os.path.dirname(pkg.LIB_PATH)
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "abi_c99.c"), "-o", str(exe), "-L", libdir, "-lglprover",
                    "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    r = subprocess.run([str(exe)], stdout=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode == 0, r.stdout
    assert "glprover" in r.stdout
