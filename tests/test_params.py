"""The three .param readers -- the oracle's, the product's and (when oracle/_ref was built in the
build container) the reference's own src/parameters.cpp -- must parse identically."""
import ctypes as C
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
DATA = os.path.join(ROOT, "tol_amd", "data")
REFSO = os.path.join(ROOT, "oracle", "_ref", "libref_params.so")
AIRCRAFT = ["tempest", "skywalker", "tempest_eric", "tempest_wences", "tempest_will"]

NASTY = "\n".join([
    "// header line is skipped",
    "6.122800\\n  // literal backslash-n as in the shipped files",
    "   -0.45\t// leading blanks",
    "1e400 // out of range: std::stod throws, line skipped",
    "abc 12 // no leading number: skipped",
    "",
    "12abc // trailing junk ignored",
    "-.5e-3/ one slash is enough",
    "0x10 // hex float is a number for stod/strtod",
    "7\r",
    "nan // parses",
    "3 4 5 // first number only",
]) + "\n"


def product_read(tolfg, path, maxn=64):
    out = np.zeros(maxn)
    cnt = tolfg.lib().tolfg_read_params(path.encode(), out.ctypes.data_as(C.POINTER(C.c_double)), maxn)
    return out[:max(cnt, 0)], cnt


def test_nasty_file(tmp_path, oracle, tolfg):
    f = tmp_path / "nasty.param"
    f.write_text(NASTY)
    vo, co = oracle.read_params(str(f))
    vp, cp = product_read(tolfg, str(f))
    assert co == cp == 8                  # 12 lines, 4 of them skipped (header, 1e400, abc, empty)
    assert np.array_equal(vo, vp, equal_nan=True)
    assert np.allclose(vo[[0, 1, 2, 3, 4, 5, 7]], [6.1228, -0.45, 12.0, -0.5e-3, 16.0, 7.0, 3.0])
    assert np.isnan(vo[6])


def test_missing_file(oracle, tolfg, tmp_path):
    with pytest.raises(FileNotFoundError):
        oracle.read_params(str(tmp_path / "nope.param"))
    _, cnt = product_read(tolfg, str(tmp_path / "nope.param"))
    assert cnt == -2        # TOLFG_ERR_PARAM


@pytest.mark.parametrize("rel", [f"aircraft/{a}.param" for a in AIRCRAFT] +
                         [f"problems/{m}/{k}.param" for m in ("S10", "G7") for k in ("gains", "limits", "snopt")])
def test_shipped_data_readers_agree(oracle, tolfg, rel):
    vo, co = oracle.read_params(os.path.join(DATA, rel))
    vp, cp = product_read(tolfg, os.path.join(DATA, rel))
    assert co == cp and np.array_equal(vo, vp)
    want = {"aircraft": 15, "gains": 5, "limits": 8, "snopt": 6}[rel.split("/")[0] if rel.startswith("aircraft") else os.path.basename(rel)[:-6]]
    assert co == want


def test_wrong_count_is_a_length_error(tmp_path, tolfg):
    """ref: std::length_error when the element count is wrong (src/parameters.cpp:45-67)."""
    root = tmp_path
    (root / "aircraft").mkdir()
    (root / "problems" / "S10").mkdir(parents=True)
    for k in ("gains", "limits", "snopt"):
        (root / "problems" / "S10" / f"{k}.param").write_text(open(os.path.join(DATA, "problems", "S10", f"{k}.param")).read())
    (root / "aircraft" / "short.param").write_text("1\n2\n3\n")
    with pytest.raises(tolfg.TolfgError) as e:
        tolfg.Problem("S10", "short", root_path=str(root))
    assert e.value.code == -2
    with pytest.raises(tolfg.TolfgError) as e:
        tolfg.Problem("S10", "absent", root_path=str(root))
    assert e.value.code == -2
    with pytest.raises(tolfg.TolfgError) as e:
        tolfg.Problem("X9", "short", root_path=str(root))
    assert e.value.code in (-1, -2)      # unknown mission: its .param files do not exist either


@pytest.mark.skipif(not os.path.exists(REFSO), reason="oracle/_ref not built (needs /root/reference: build container only)")
@pytest.mark.parametrize("root", [DATA, "/root/reference"])
def test_against_the_reference_reader(oracle, root):
    """The reference's own reader (compiled in place from src/parameters.cpp) on this repo's data
    files and, in the build container, on the reference's files: same values as the oracle reads."""
    if not os.path.isdir(os.path.join(root, "aircraft")):
        pytest.skip(f"{root} absent")
    R = C.CDLL(REFSO)
    dp = C.POINTER(C.c_double)
    for fn in (R.ref_aircraft, R.ref_gain, R.ref_limit, R.ref_snopt):
        fn.argtypes = [C.c_char_p, C.c_char_p, dp]
        fn.restype = C.c_int
    rootb = (root.rstrip("/") + "/").encode()
    for a in AIRCRAFT:
        out = np.zeros(15)
        assert R.ref_aircraft(a.encode(), rootb, out.ctypes.data_as(dp)) == 0
        mine, cnt = oracle.read_params(os.path.join(root, "aircraft", a + ".param"))
        assert cnt == 15
        mine = mine.copy()
        mine[[8, 11, 12]] = mine[[8, 11, 12]] * np.pi / 180.0    # as the reference: v*M_PI/180.0
        assert np.array_equal(out, mine)
    for m in ("S10", "G7"):
        for fn, kind, cntw in ((R.ref_gain, "gains", 5), (R.ref_limit, "limits", 8), (R.ref_snopt, "snopt", 6)):
            out = np.zeros(8)
            assert fn(m.encode(), rootb, out.ctypes.data_as(dp)) == 0
            mine, cnt = oracle.read_params(os.path.join(root, "problems", m, kind + ".param"))
            assert cnt == cntw
            if kind == "limits":      # the reference stores xmin xmax ymin ymax zmin zmax in file order
                assert np.array_equal(out[:8], mine)
            else:
                assert np.array_equal(out[:cntw], mine)
    # wrong element count -> the reference throws, the shim reports -1
    out = np.zeros(15)
    assert R.ref_aircraft(b"does_not_exist", rootb, out.ctypes.data_as(dp)) == -1


@pytest.mark.skipif(not os.path.exists(REFSO), reason="oracle/_ref not built (needs /root/reference: build container only)")
def test_reference_reader_on_a_nasty_file(tmp_path, oracle):
    """Junk lines, CRs, literal backslash-n, out-of-range and hex numbers: the reference's
    stod-based reader and the oracle's strtod-based one keep and drop the same lines."""
    (tmp_path / "aircraft").mkdir()
    lines = ["// title", "1\\n // a", "2\r", "junk // dropped", "1e999 // dropped", "3.5e0abc", "  4", "5/x", "6 // x", "7", "8", "9",
             "0x1p3 // hex = 8", "11", "12", "", "13", "14", "-15"]
    (tmp_path / "aircraft" / "nasty.param").write_text("\n".join(lines) + "\n")
    mine, cnt = oracle.read_params(str(tmp_path / "aircraft" / "nasty.param"))
    assert cnt == 15
    R = C.CDLL(REFSO)
    dp = C.POINTER(C.c_double)
    R.ref_aircraft.argtypes = [C.c_char_p, C.c_char_p, dp]
    R.ref_aircraft.restype = C.c_int
    out = np.zeros(15)
    assert R.ref_aircraft(b"nasty", (str(tmp_path) + "/").encode(), out.ctypes.data_as(dp)) == 0
    mine = mine.copy()
    mine[[8, 11, 12]] = mine[[8, 11, 12]] * np.pi / 180.0
    assert np.array_equal(out, mine)


_TOKENS = ["1", "-2.5e3", "+.5", "1.", "0x1p3", "0x10", "inf", "-inf", "nan", "1e999", "-1e-999", " 3", "\t4", "5abc", "6/7", "7 // c", "8\\n", "9\r",
           "", "abc", "/ 5", "--1", "1e", "1e+", ".", "-", "+", "0x", "1_000", "1,5", "  ", "// 3", "3 4", "1e5e5", ".e1", "00012", "1d3", "infinity x",
           "nanx", "-0", "1e-320", "\r", "12 / 13 / 14", "\\n", "0.1e+2/"]


def _random_file(rng, n_lines):
    return "\n".join(_TOKENS[int(i)] for i in rng.integers(0, len(_TOKENS), n_lines)) + "\n"


@pytest.mark.parametrize("seed", range(8))
def test_readers_agree_on_random_line_soups(tmp_path, oracle, tolfg, seed):
    """Seeded random files over a vocabulary of well- and ill-formed lines: the product's reader and the oracle's keep and drop the
    same lines and read the same values, and -- where oracle/_ref exists (build container) -- so does the reference's own reader
    (compiled in place from src/parameters.cpp:14-34): a file that yields exactly 15 values must give the reference's aircraft
    constructor the same 15 numbers, any other count must make it throw (src/parameters.cpp:45-67)."""
    rng = np.random.default_rng(4400 + seed)
    R = None
    if os.path.exists(REFSO):
        R = C.CDLL(REFSO)
        R.ref_aircraft.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_double)]
        R.ref_aircraft.restype = C.c_int
    (tmp_path / "aircraft").mkdir()
    hits = 0
    for k in range(60):
        text = _random_file(rng, int(rng.integers(10, 40)))
        f = tmp_path / "aircraft" / f"soup{k}.param"
        f.write_bytes(text.encode())
        vo, co = oracle.read_params(str(f))
        vp, cp = product_read(tolfg, str(f))
        assert co == cp, (seed, k, text)
        assert np.array_equal(vo, vp, equal_nan=True), (seed, k, text)
        if R is not None:
            out = np.zeros(15)
            rc = R.ref_aircraft(f"soup{k}".encode(), (str(tmp_path) + "/").encode(), out.ctypes.data_as(C.POINTER(C.c_double)))
            if co == 15:
                hits += 1
                want = vo.copy()
                want[[8, 11, 12]] = want[[8, 11, 12]] * np.pi / 180.0
                assert rc == 0 and np.array_equal(out, want, equal_nan=True), (seed, k, text)
            else:
                assert rc == -1, (seed, k, co, text)
    # a count of exactly 15 is rare in a random soup, so also grow files line by line until they hold exactly 15 values
    for k in range(12):
        lines = []
        while True:
            lines.append(_TOKENS[int(rng.integers(0, len(_TOKENS)))])
            f = tmp_path / "aircraft" / f"grown{k}.param"
            f.write_bytes(("\n".join(lines) + "\n").encode())
            vo, co = oracle.read_params(str(f))
            if co >= 15:
                break
        vp, cp = product_read(tolfg, str(f))
        assert co == cp == 15 and np.array_equal(vo, vp, equal_nan=True), (seed, k, lines)
        if R is not None:
            out = np.zeros(15)
            assert R.ref_aircraft(f"grown{k}".encode(), (str(tmp_path) + "/").encode(), out.ctypes.data_as(C.POINTER(C.c_double))) == 0, (seed, k, lines)
            want = vo.copy()
            want[[8, 11, 12]] = want[[8, 11, 12]] * np.pi / 180.0
            assert np.array_equal(out, want, equal_nan=True), (seed, k, lines)
