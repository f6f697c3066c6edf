"""include/tolfg.h is plain C: the C99 example compiles with gcc -pedantic (CPU) and, on the GPU box,
runs the batched ABI without Python or C++ in the process; its objectives match the oracle."""
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(ROOT, "examples", "batch_montecarlo.c")


def build(tolfg, out):
    libdir = os.path.dirname(tolfg.lib_path())
    subprocess.run(["gcc", "-std=c99", "-O2", "-Wall", "-Wextra", "-pedantic", "-Werror", "-Wno-unused-parameter",
                    "-I", os.path.join(ROOT, "include"), "-isystem", "/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", SRC,
                    "-o", out, "-L", libdir, "-ltolfg", "-L", "/opt/rocm/lib", "-lamdhip64", "-lm",
                    f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"], check=True)


def test_header_is_plain_c(tolfg, tmp_path):
    build(tolfg, str(tmp_path / "mc"))


@pytest.mark.gpu
def test_c_example_runs(tolfg, oracle, tmp_path):
    exe = str(tmp_path / "mc")
    build(tolfg, exe)
    B = 33
    out = subprocess.run([exe, str(B)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    w = out.stdout.split()
    first, last = float(w[w.index("first") + 1]), float(w[w.index("last") + 1])
    for t, got in ((0, first), (B - 1, last)):
        o = oracle.Problem("S10", "tempest", N=200, Vref=5.0 * t / (B - 1), href=10.0, start=(0.0, 0.0, -50.0))
        want = o.eval(o.x0(), needG=False)[0][0]
        assert got == pytest.approx(want, rel=1e-12)
