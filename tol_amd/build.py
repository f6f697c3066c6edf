"""Build tol_amd/lib/libtolfg.so (the product) and libtolfg_measure.so (the same sources with the measurement variables of
csrc/knobs.h compiled in) with hipcc for gfx950 -- in-tree, so the .so files travel with the repo."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libtolfg.so")
MEASURE_LIB = os.path.join(HERE, "lib", "libtolfg_measure.so")


def build(force=False):
    args = ["make", "-s", "-j4", "-C", CSRC, "all"]      # kernels.hip is three translation units: they build in parallel
    if force:
        args.append("-B")
    subprocess.run(args, check=True)
    for lib in (LIB, MEASURE_LIB):
        if not os.path.exists(lib):
            raise RuntimeError("hipcc did not produce " + lib)
    return LIB


if __name__ == "__main__":
    print(build())
