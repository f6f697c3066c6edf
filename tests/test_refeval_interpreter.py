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


# ---- round 3: constructors, try / catch around an absent library, class declarations, big work arrays

HDR = r'''
class Base {
public:
    virtual ~Base();
    void run(double x[]);
    bool flag;
protected:
    Base(Args& a);
    lib::Connection db;
    const double g = 9.81;
    int mode;
    double a1, a2 = 2.5, a3;
    std::string name;
    std::vector<double> w, w2;
    std::vector<std::vector<int>> nest;
    int *idx;
    double *buf;
    struct Doc { public: Doc(double x) : x(x) {} double x; };
    int count;
};
class Derived : public Base {
public:
    Derived(Args& a);
protected:
    void fill();
    double extra;
};
'''
CTORS = r'''
Base::Base(Args& a): par(a.name), other(a.k)
{
    cout << "building " << a.name << endl;
    flag = true;
    name = a.name;
    a1 = max(a.k, 1.0);
    try {
        cout << "connecting...";
        lib::client::initialize();
        db.connect("host:1");
        mode = 3;
    }
    catch (exception& e) {
        cout << "failure" << endl;
        mode = 1;
    }
    if (a.name == "two") { a3 = 2; }
    count = a.n * (a.n + 1);
    w.resize(a.n + 1);
    idx = new int[count];
    buf = new double[a.n];
    w2.resize(count);
}
Derived::Derived(Args& a) : Base(a) {
    fill();
}
void Derived::fill() { for (int i = 0; i < count; i++) { idx[i] = i; } extra = a1 + a2 + w[0]; }
int Base::direct() { lib::client::initialize(); return 1; }
'''


def test_class_declaration_gives_every_member_its_cpp_default():
    m = cinterp.declare_members(HDR, "Base")
    assert m["flag"] == 0 and m["mode"] == 0 and m["count"] == 0 and m["g"] == 9.81 and m["a2"] == 2.5 and m["name"] == ""
    assert math.isnan(m["a1"]) and math.isnan(m["a3"])                       # uninitialised doubles
    assert m["w"] == [] and m["w2"] == [] and m["nest"] == [] and m["idx"] == [] and m["buf"] == []
    assert isinstance(m["db"], cinterp.Unavailable)
    assert "run" not in m and "Doc" not in m and "x" not in m and "Base" not in m
    assert cinterp.declare_members(HDR, "Derived") == pytest.approx({"extra": float("nan")}, nan_ok=True)
    with pytest.raises(NameError):
        cinterp.declare_members(HDR, "Missing")


def test_constructors_run_base_first_and_an_absent_library_lands_in_the_catch_block():
    it2 = cinterp.Interp([CTORS])
    o = NS(_classes=["Derived", "Base"])
    for cls in ("Base", "Derived"):
        for k, v in cinterp.declare_members(HDR, cls).items():
            setattr(o, k, v)
    seen = []
    it2.construct(o, "Derived", NS(name="two", k=0.25, n=3), between=lambda ob: seen.append((ob.mode, ob.count)))
    assert seen == [(1, 12)]                        # the hook ran after Base::Base (catch block taken), before Derived's body
    assert o.flag == 1 and o.name == "two" and o.mode == 1 and o.a1 == 1.0 and o.a3 == 2.0
    assert o.w == [0.0] * 4 and o.w2 == [0.0] * 12 and o.idx == list(range(12)) and len(o.buf) == 3 and math.isnan(o.buf[0])
    assert o.extra == 1.0 + 2.5 + 0.0
    with pytest.raises(cinterp.ExternalUnavailable):        # outside a try block the absent library is an error, not a skip
        it2.call(o, "direct")


def test_huge_work_arrays_are_held_sparsely():
    it2 = cinterp.Interp(["void K::big(int n) { a = new int[n]; v.resize(n); a[n - 1] = 7; v[5] = 2.5; }"])
    o = NS(_classes=["K"], a=[], v=[])
    it2.call(o, "big", cinterp.BIG + 10)
    assert isinstance(o.a, cinterp.BigArray) and len(o.a) == cinterp.BIG + 10 and o.a[cinterp.BIG + 9] == 7 and o.a[3] == 0
    assert isinstance(o.v, cinterp.BigArray) and o.v[5] == 2.5 and o.v[6] == 0.0 and o.v[4:7] == [0.0, 2.5, 0.0]
