"""The C-subset interpreter behind tests/golden/ref_eval_vectors.npz (tools/refeval/cinterp.py) evaluates C the
way C does: these are self-written snippets (no reference text) exercising the semantics the reference's
functions rely on -- integer division and remainder, int <-> double conversion at assignments and calls,
switch fall-through and default, uninitialised scalars, array initialisers, member state, file output."""
import math
import os
import sys
from types import SimpleNamespace as NS

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "refeval"))
import cinterp  # noqa: E402

SRC = r'''
int K::idiv(int a, int b) { return a / b; }
int K::imod(int a, int b) { return a % b; }
double K::mixed(int a, double b) { int t; t = b; double h = a / 2; return t + h + a / 2.0; }
int K::trunc_at_call(double v) { return idiv(v, 1); }
double K::sw(int k) {
    double r = 0;
    switch (k) {
    case 1: r = r + 1;
    case 2: r = r + 10; break;
    case 3: { r = 100; break; }
    default: r = -1; break;
    }
    return r;
}
double K::uninit(int k) { double g; if (k == 1) { g = 5.0; } return g; }
double K::arrays(int n) {
    double t[6] = {1.5, 2.5};
    int idx[3];
    double s = 0;
    for (int i = 0; i < n; i++) { idx[i] = i * 2; s += t[i] + idx[i]; }
    int j = 0;
    while (j < 3) { j++; if (j == 2) continue; s += 0.25; }
    return s + t[5];
}
void K::members(double x[]) {
    counter = counter + 1.9;          // int member: truncates
    total += x[ind + 1] * gain.k;
    flags[counter] = 7.7;             // vector<double> element
    FILE* fp;
    fp = fopen("out.txt", "w");
    for (int i = 0; i < 2; i++) { fprintf(fp, "%.3f ", x[i]); }
    fprintf(fp, "%d\n", counter);
    fclose(fp);
    cout << "ignored" << endl;
    auto t0 = std::chrono::system_clock::now();
}
double K::logic(double a, double b) { return ((a > 1) && (b <= 2)) || !(a == a) ? cos(0.0) + M_PI : pow(a, 2) - fabs(b); }
'''


@pytest.fixture()
def it():
    return cinterp.Interp([SRC])


def obj():
    return NS(_classes=["K"], counter=0, total=0.0, ind=1, gain=NS(k=2.0), flags=[0.0, 0.0, 0.0])


def test_integer_division_and_remainder_follow_c(it):
    o = obj()
    assert it.call(o, "idiv", 7, 2) == 3 and it.call(o, "idiv", -7, 2) == -3 and it.call(o, "idiv", -1, 8) == 0
    assert it.call(o, "imod", 7, 3) == 1 and it.call(o, "imod", -7, 3) == -1 and it.call(o, "imod", 16, 8) == 0
    assert it.call(o, "trunc_at_call", 3.99) == 3 and it.call(o, "trunc_at_call", -3.99) == -3


def test_conversions_at_assignment(it):
    # t = 2.9 -> 2 ; h = 7 / 2 -> 3 (integer division, then to double) ; 7 / 2.0 -> 3.5
    assert it.call(obj(), "mixed", 7, 2.9) == 2 + 3.0 + 3.5


def test_switch_falls_through_and_defaults(it):
    o = obj()
    assert [it.call(o, "sw", k) for k in (1, 2, 3, 9)] == [11.0, 10.0, 100.0, -1.0]


def test_uninitialised_scalar_reads_as_nan(it):
    assert it.call(obj(), "uninit", 1) == 5.0
    assert math.isnan(it.call(obj(), "uninit", 0))


def test_arrays_loops_and_zero_fill(it):
    # t = {1.5, 2.5, 0, 0, 0, 0}; i = 0..2: (1.5+0) + (2.5+2) + (0+4) = 10 ; the while adds 0.25 twice ; + t[5] = 0
    assert it.call(obj(), "arrays", 3) == 10.0 + 0.5


def test_member_state_and_file_output(it):
    o = obj()
    it.call(o, "members", [1.0, 2.0, 3.0])
    assert o.counter == 1 and o.total == 6.0 and o.flags == [0.0, 7.7, 0.0]
    assert it.output.files["out.txt"] == "1.000 2.000 1\n"
    with pytest.raises(IndexError):
        it.call(NS(_classes=["K"], counter=5, total=0.0, ind=1, gain=NS(k=1.0), flags=[0.0]), "members", [1.0, 2.0, 3.0])


def test_logic_ternary_and_libm(it):
    o = obj()
    assert it.call(o, "logic", 2.0, 1.0) == 1.0 + math.pi
    assert it.call(o, "logic", 0.5, -3.0) == 0.25 - 3.0
    assert it.call(o, "logic", float("nan"), 0.0) == 1.0 + math.pi
