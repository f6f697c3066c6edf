"""Build tol_amd/lib/libtolfg.so with hipcc for gfx950 (in-tree, so the .so travels with the repo)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libtolfg.so")


def build(force=False):
    args = ["make", "-s", "-j4", "-C", CSRC]      # kernels.hip is three translation units: they build in parallel
    if force:
        args.append("-B")
    subprocess.run(args, check=True)
    if not os.path.exists(LIB):
        raise RuntimeError("hipcc did not produce " + LIB)
    return LIB


if __name__ == "__main__":
    print(build())
