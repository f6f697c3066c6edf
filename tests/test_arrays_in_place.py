"""Arrays used in place (include/tolfg.h): the kernel reads x and writes F, G in the caller's memory only under an
explicit contract -- tolfg_register_arrays(), or tolfg_config.persistent_arrays -- never because an address was seen
twice; tolfg_forget_arrays() ends the contract before the caller lets go of its arrays.  And a lost objective partial
surfaces as an error code (tolfg_batch_status / *Status = -2), not as a silent NaN."""
import gc

import numpy as np
import pytest

from helpers import assert_close


def test_forget_and_count_need_no_gpu(tolfg):
    p = tolfg.Problem("S10", "tempest", ts=20)
    assert p.registered_arrays() == 0
    p.forget_arrays()                       # nothing registered: a no-op, no GPU touched
    assert p.registered_arrays() == 0
    import torch
    if not torch.cuda.is_available():       # registering pins memory for a device: fails loudly without one
        with pytest.raises(tolfg.TolfgError):
            p.register_arrays(F=np.zeros(p.neF))
    p.close()


@pytest.mark.gpu
def test_fresh_arrays_are_never_pinned(tolfg, oracle):
    """The default: a caller that allocates new F and G for every call (host.Problem.define_fg does) -- addresses
    repeat, arrays do not persist -- must never have anything registered."""
    p = tolfg.Problem("S10", "tempest", ts=64)
    o = oracle.Problem("S10", "tempest", N=64)
    x = oracle.perturbed(o, 3)
    Fo, Go = o.eval(x)
    for _ in range(40):
        F, G, st = p.define_fg(x.copy())
        assert st == 1
        assert p.registered_arrays() == 0
        del F, G
        gc.collect()
    F, G, st = p.define_fg(x)
    assert_close(F, Fo, what="F")
    assert_close(G, Go, mask=o.undefined_mask(), what="G")
    p.close()


@pytest.mark.gpu
@pytest.mark.parametrize("how", ["explicit", "persistent_arrays"])
def test_in_place_contract_and_reallocation(tolfg, oracle, how):
    """Arrays kept across calls are used in place; after forget_arrays() the caller may free and re-allocate them (at the
    same address or elsewhere) and the next calls land in the NEW arrays."""
    N = 200
    p = tolfg.Problem("S10", "tempest", ts=N, persistent_arrays=(how == "persistent_arrays"))
    o = oracle.Problem("S10", "tempest", N=N)
    mask = o.undefined_mask()
    xs = [oracle.perturbed(o, s) for s in (1, 2, 3)]
    refs = [o.eval(x) for x in xs]
    x = xs[0].copy()
    F, G = np.zeros(p.neF), np.zeros(p.neG)
    if how == "explicit":
        p.register_arrays(x, F, G)
        assert p.registered_arrays() == 3
    for k in (0, 1, 2, 1):
        x[:] = xs[k]
        _, _, st = p.define_fg(x, F=F, G=G)
        assert st == 1
        assert_close(F, refs[k][0], what=f"{how} F pt{k}")
        assert_close(G, refs[k][1], mask=mask, what=f"{how} G pt{k}")
    assert p.registered_arrays() == 3          # x, F, G: seen twice in a row, or registered explicitly
    # the caller lets go of its arrays: contract ended first
    p.forget_arrays()
    assert p.registered_arrays() == 0
    addr = (F.ctypes.data, G.ctypes.data)
    del F, G
    gc.collect()
    for _ in range(3):                          # new arrays, possibly at the old addresses
        F, G = np.full(p.neF, 7.0), np.full(p.neG, 7.0)
        for k in (2, 0):
            x[:] = xs[k]
            _, _, st = p.define_fg(x, F=F, G=G)
            assert st == 1
            assert_close(F, refs[k][0], what=f"{how} F after re-allocation")
            assert_close(G, refs[k][1], mask=mask, what=f"{how} G after re-allocation")
        p.forget_arrays()
        del F, G
        gc.collect()
    assert isinstance(addr, tuple)
    p.close()


@pytest.mark.gpu
def test_time_callback_leaves_nothing_pinned(tolfg):
    p = tolfg.Problem("S10", "skywalker", ts=2000)
    x = p.x0()
    for _ in range(2):
        us, F, G = p.time_callback(x, 5, warm=3)
        assert us > 0 and np.isfinite(F).all() and F[0] != 0.0
        assert p.registered_arrays() == 0
    p.close()


MARKER = np.array([0xFFFBADADFFFBADAD], dtype=np.uint64).view(np.float64)[0]     # kernels.hip kEmptySlot


@pytest.mark.gpu
def test_lost_partial_is_reported_by_the_batch(tolfg):
    """A thrust value that IS the empty-slot marker (a NaN with that payload) makes its tile's objective partial look
    like an empty slot for good: the launch must end by itself, and the host must be told."""
    import os
    import torch
    if os.environ.get("TOLFG_FUSED") == "0":
        pytest.skip("the two-launch form (a measurement override) has no polled slots: nothing can be lost there")
    N, B = 200, 12               # more than 8 trajectories: the tile-per-workgroup path with its polled partial slots
    bt = tolfg.Batch("S10", ["tempest"], ts=N)
    bt.set_trajectories([tolfg.Trajectory() for _ in range(B)])
    dX, dF, dG = bt.alloc(B)
    bt.x0_device(dX)
    bt.eval(dX, dF, dG)
    torch.cuda.synchronize()
    bt.status()                                  # healthy
    good = dF[:, 0].clone()
    bad = dX.clone()
    bad[3, 1 + 11 * 7 + 10] = float(MARKER)      # thrust of node 7 of trajectory 3
    assert bad[3, 1 + 11 * 7 + 10].view(torch.int64).item() == np.float64(MARKER).view(np.int64)
    bt.eval(bad, dF, dG)
    torch.cuda.synchronize()                     # the launch ends (bounded wait), no hang
    ok = [t for t in range(B) if t != 3]
    assert torch.isnan(dF[3, 0]) and torch.equal(dF[ok, 0], good[ok])
    with pytest.raises(tolfg.TolfgError) as e:
        bt.status()
    assert e.value.code == tolfg.capi.ERR_HIP and "partial" in str(e.value)
    bt.status()                                  # reported once, then clear
    # the workspace was left ready: the next evaluation of good inputs is right again
    bt.eval(dX, dF, dG)
    torch.cuda.synchronize()
    bt.status()
    assert torch.equal(dF[:, 0], good)
    # an unread condition stops the next evaluation instead of being overwritten
    bt.eval(bad, dF, dG)
    torch.cuda.synchronize()
    with pytest.raises(tolfg.TolfgError):
        bt.eval(dX, dF, dG)
    bt.eval(dX, dF, dG)
    torch.cuda.synchronize()
    bt.status()
    bt.close()


@pytest.mark.gpu
def test_lost_partial_sets_status_in_the_callback(tolfg, capfd):
    """ts = 300 takes the tile-per-workgroup path (the one-workgroup kernel sums in LDS and cannot lose anything)."""
    N = 300
    p = tolfg.Problem("S10", "tempest", ts=N)
    x = p.x0()
    F, G, st = p.define_fg(x)
    assert st == 1 and np.isfinite(F).all()
    xb = x.copy()
    xb[1 + 11 * 100 + 10] = MARKER
    F, G, st = p.define_fg(xb)
    assert st == -2 and np.isnan(F[0])
    assert "partial" in capfd.readouterr().err
    F, G, st = p.define_fg(x)                    # and the problem is usable again
    assert st == 1 and np.isfinite(F).all()
    p.close()
