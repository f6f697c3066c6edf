"""Shared helpers for the parity tests."""
import numpy as np

# fp64 bar: |hip - oracle| <= 1e-12 * (1 + |oracle|) on F and every defined G
# entry.  Bit equality is not attainable: the kernel fuses multiply-adds, uses reciprocals for 1/m,
# 1/Va, 1/cos(gam) and sums the objective as a butterfly, the oracle does none of that.
RTOL64 = 1e-12


def assert_close(got, ref, tol=RTOL64, mask=None, what=""):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    err = np.abs(got - ref) / (1.0 + np.abs(ref))
    if mask is not None:
        err = np.where(mask, 0.0, err)
    bad = ~np.isfinite(got)
    if mask is not None:
        bad &= ~mask
    assert not bad.any(), f"{what}: non-finite output at {np.flatnonzero(bad)[:8]}"
    worst = int(np.argmax(err))
    assert err[worst] <= tol, f"{what}: worst scaled error {err[worst]:.3e} at {worst}: {got.flat[worst]!r} vs {ref.flat[worst]!r}"
    return float(err.max()) if err.size else 0.0


def random_wind_table(N, seed, scale=0.3):
    """12 x (N+1) ENU wind values exercising every gradient term (oracle 'wind injection')."""
    rng = np.random.default_rng(seed)
    w = rng.uniform(-scale, scale, (12, N + 1))
    w[:3] *= 10.0
    return w
