#!/usr/bin/env python3
"""Regenerate tol_amd/data/**.param from the reference's parameter VALUES.

Runs only in the build container (needs /root/reference).  The reference's .param files are read as
text with the reference reader's semantics (src/parameters.cpp:14-34: text before the first '/',
leading float, unparsable lines skipped); only the numeric tokens are kept.  The files written here
carry this repo's own field labels, one `value // label` per line, which both this repo's reader
and the reference's reader parse to the same numbers (tests/test_params.py checks both).
"""
import os
import re
import sys

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tol_amd", "data")

LABELS = {
    "aircraft": ["mm: mass [kg]", "b: span [m]", "SS: wing area [m^2]", "ee: Oswald factor [-]",
                 "AR: aspect ratio [-]", "Cd0: zero-lift drag [-]", "CLmin [-]", "CLmax [-]",
                 "phimax: bank limit [deg]", "Vamin [m/s]", "Vamax [m/s]", "gammamax: climb limit [deg]",
                 "phidotmax: roll-rate limit [deg/s]", "Tmin [N]", "Tmax [N]"],
    "gains": ["kT: thrust^2 weight", "kp: position weight", "kv: speed weight", "ka: angle weight (unused)",
              "kdt: time weight"],
    "limits": ["dtmin [s]", "dtmax [s]", "xmin [m]", "xmax [m]", "ymin [m]", "ymax [m]", "zmin [m]", "zmax [m]"],
    "snopt": ["ts: time segments", "numinp: variables per node", "numstates: states per node",
              "numbounds: boundary rows", "opt_tol: major optimality tolerance",
              "feas_tol: major feasibility tolerance"],
}

NUM = re.compile(r"^\s*([-+]?(?:\d+\.?\d*|\.\d+)(?:[eE][-+]?\d+)?)")


def tokens(path):
    out = []
    with open(path, "r", errors="replace") as fh:
        for line in fh.read().split("\n"):
            head = line.split("/", 1)[0]
            m = NUM.match(head)
            if m:
                out.append(m.group(1))
    return out


def emit(src, dst, kind, title):
    vals = tokens(src)
    labels = LABELS[kind]
    if len(vals) != len(labels):
        sys.exit(f"{src}: {len(vals)} values, expected {len(labels)}")
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    with open(dst, "w") as fh:
        fh.write(f"// {title} -- values as in lingaqing/tol, labels by tolfg-mi355x\n")
        for v, lab in zip(vals, labels):
            fh.write(f"{v:<12s}// {lab}\n")


def main():
    for f in sorted(os.listdir(os.path.join(REF, "aircraft"))):
        if f.endswith(".param"):
            emit(os.path.join(REF, "aircraft", f), os.path.join(OUT, "aircraft", f), "aircraft",
                 f"airframe {f[:-6]}")
    for m in ("S10", "G7"):
        for kind in ("gains", "limits", "snopt"):
            emit(os.path.join(REF, "problems", m, kind + ".param"),
                 os.path.join(OUT, "problems", m, kind + ".param"), kind, f"mission {m} {kind}")


if __name__ == "__main__":
    main()
