"""One and several RK4 steps on the device (tpsrhs_rk4_step) against the oracle's
restatement of M2ulPhyS::solveStep's integrator call + Check_NAN + Check_Undershoot."""
import numpy as np
import pytest

from oracle_lib import Oracle
from parity_util import rel_maxnorm
from tps_amd import capi, cases

pytestmark = pytest.mark.gpu


def _run(c, U, dt, nsteps):
    import torch
    from tps_amd.rhs_operator import RHSoperator

    op = RHSoperator(c.mesh, c.disc, c.physics, c.bcs)
    x = torch.tensor(np.ascontiguousarray(U).ravel(), dtype=torch.float64, device=op.device)
    t = 0.0
    for _ in range(nsteps):
        t = op.rk4_step(x, t, dt, want_max_char_speed=True, want_nan_count=True)
    torch.cuda.synchronize()
    out = x.cpu().numpy().reshape(U.shape), t, op.max_char_speed, op.nan_count
    op.close()
    return out


@pytest.mark.parametrize("kind", ["dry_air", "argon_2T"])
def test_rk4_steps_match_oracle(kind):
    if kind == "dry_air":
        c = cases.cyl3d(4, 12, 3, 2, capi.NS, capi.VISC_ISOTH)
        c.physics.dry_air.visc_mult = 100.0
        U = c.state(seed=2)
    else:
        c = cases.argon_cyl3d(4, 12, 3, 2, True, capi.CONSTANT, "arrhenius", capi.VISC_ISOTH)
        U = c.state(seed=2, amp=0.01)
    o = Oracle(c.mesh, c.disc, c.physics, c.bcs)
    y0 = o.mult(U)
    # a stable explicit step: a tenth of the fastest local time scale of the residual (the electron energy
    # exchange of the two-temperature plasma is much stiffer than the acoustic CFL limit)
    dt = 0.1 / (np.abs(y0) / np.maximum(np.abs(U), 1e-300 + 1e-6 * np.abs(U).max(axis=1, keepdims=True))).max()
    nsteps = 3
    ref, t = U.copy(), 0.0
    for _ in range(nsteps):
        ref, t, speed, bad = o.rk4_step(ref, t, dt)
    got, tg, gspeed, gbad = _run(c, U, dt, nsteps)
    assert tg == pytest.approx(t, rel=1e-15) and gbad == bad == 0
    # the state moves by O(dt |f|): measure the difference against that increment
    incr = np.abs(ref - U).reshape(U.shape[0], -1).max(axis=1)
    err = np.abs(got - ref).reshape(U.shape[0], -1).max(axis=1)
    print("increment", incr, "difference", err)
    assert (err <= 1e-9 * incr + 1e-15 * np.abs(U).reshape(U.shape[0], -1).max(axis=1)).all()
    assert rel_maxnorm(got, ref).max() < 1e-13
    assert gspeed == pytest.approx(speed, rel=1e-12)


def test_rk4_counts_nans_and_clamps_species():
    c = cases.argon_cyl3d(4, 12, 3, 1, False, capi.CONSTANT, None, capi.VISC_ISOTH)
    U = c.state(seed=3, amp=0.005)
    U[0, 5] = np.nan  # one bad density entry poisons its element
    got, _, _, bad = _run(c, U, 1e-9, 1)
    # the census (Check_NAN) runs before the clamp (Check_Undershoot), and max(NaN, 0) = 0 in the species rows
    assert bad >= np.isnan(got).sum() > 0
    assert not np.isnan(got[5]).any() and np.isnan(got[:5]).sum() == np.isnan(got).sum()


@pytest.mark.parametrize("constant_dt", [True, False])
def test_advance_keeps_the_time_loop_on_the_device(constant_dt):
    """tpsrhs_advance: several solveStep's with dt (CFL controlled or constant), time and NaN census on the device,
    against the oracle's host loop."""
    import torch
    from tps_amd.rhs_operator import RHSoperator

    c = cases.cyl3d(4, 12, 3, 2, capi.NS, capi.VISC_ISOTH)
    c.physics.dry_air.visc_mult = 100.0
    # a non-reflecting outlet on top: its boundary state integrates with the device-side dt too
    c.bcs[1] = capi.make_bc(2, capi.OUTLET, capi.SUB_P_NR, [101000.0, 0, 0, 0, 0.0, 0.0, 1.0, 0.0])
    U = c.state(seed=2)
    o = Oracle(c.mesh, c.disc, c.physics, c.bcs)
    dt0, cfl, hmin, nsteps = 2.0e-5, 0.12, 0.05, 4
    ref, t, dt, bad = o.advance(U, 0.0, dt0, nsteps, constant_dt, cfl, hmin)
    op = RHSoperator(c.mesh, c.disc, c.physics, c.bcs)
    x = torch.tensor(np.ascontiguousarray(U).ravel(), dtype=torch.float64, device=op.device)
    tg, dtg, gbad = op.advance(x, 0.0, dt0, nsteps, constant_dt, cfl, hmin)
    got = x.cpu().numpy().reshape(U.shape)
    op.close()
    print("time", tg, t, "next dt", dtg, dt)
    assert gbad == bad == 0
    assert tg == pytest.approx(t, rel=1e-13) and dtg == pytest.approx(dt, rel=1e-12)
    assert (dtg == dt0) == constant_dt
    assert rel_maxnorm(got, ref).max() < 1e-13


def test_advance_replays_a_captured_step(monkeypatch):
    """On a capturable stream tpsrhs_advance turns one RK4 step into a hipGraph and replays it: same numbers as the
    plain launch loop, bit for bit, and as the oracle."""
    import time

    import torch
    from tps_amd.rhs_operator import RHSoperator

    c = cases.cyl3d(4, 12, 3, 3, capi.NS, capi.VISC_ISOTH)
    c.physics.dry_air.visc_mult = 100.0
    c.bcs[1] = capi.make_bc(2, capi.OUTLET, capi.SUB_P_NR, [101000.0, 0, 0, 0, 0.0, 0.0, 1.0, 0.0])
    U = c.state(seed=2)
    forcing = capi.make_forcing(pressure_gradient=(2.0, 0.0, -1.0))
    dt0, cfl, hmin, nsteps = 2.0e-5, 0.12, 0.05, 6
    o = Oracle(c.mesh, c.disc, c.physics, c.bcs)
    o.set_forcing(forcing)
    ref, t, dt, _ = o.advance(U, 0.0, dt0, nsteps, False, cfl, hmin)
    results, timings = {}, {}
    for mode in ("1", "0"):
        monkeypatch.setenv("TPSRHS_GRAPH", mode)
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            op = RHSoperator(c.mesh, c.disc, c.physics, c.bcs, stream=side)
            op.setForcing(forcing)
            x = torch.tensor(np.ascontiguousarray(U).ravel(), dtype=torch.float64, device=op.device)
            tg, dtg, bad = op.advance(x, 0.0, dt0, nsteps, False, cfl, hmin)
            results[mode] = (x.cpu().numpy().reshape(U.shape), tg, dtg, bad)
            # launch-bound regime: many steps of this 144-element mesh, with and without the graph
            side.synchronize()
            t0 = time.perf_counter()
            op.advance(x, 0.0, 1e-12, 200, True)
            timings[mode] = (time.perf_counter() - t0) / 200
            op.close()
    print("per RK4 step: graph %.1f us, launch loop %.1f us" % (1e6 * timings["1"], 1e6 * timings["0"]))
    g, p = results["1"], results["0"]
    assert np.array_equal(g[0], p[0]) and g[1:] == p[1:]
    assert g[1] == pytest.approx(t, rel=1e-13) and g[2] == pytest.approx(dt, rel=1e-12) and g[3] == 0
    assert rel_maxnorm(g[0], ref).max() < 1e-13


def test_time_loop_is_fourth_order_in_dt():
    """RK4 through tpsrhs_advance: halving dt divides the time-integration error by 16 (the spatial operator is the
    same in all runs, so the differences between them are purely temporal)."""
    import torch
    from tps_amd import meshgen
    from tps_amd.rhs_operator import RHSoperator, node_coordinates

    mesh = meshgen.box_hex(3, 3, 3, lengths=(1.0, 0.8, 1.2), warp=0.05)
    ph = capi.dry_air_physics(capi.NS, visc_mult=2.0e3)
    U = cases.dry_air_state(node_coordinates(mesh, 2), seed=5)
    op = RHSoperator(mesh, capi.Disc(2, 0, 0, 0, 0), ph, [])
    t_end, base = 4.0e-4, 10

    def run(nsteps):
        x = torch.tensor(np.ascontiguousarray(U).ravel(), dtype=torch.float64, device=op.device)
        t, _, bad = op.advance(x, 0.0, t_end / nsteps, nsteps, True)
        assert bad == 0 and t == pytest.approx(t_end, rel=1e-12)
        return x.cpu().numpy()

    ref = run(16 * base)
    e1, e2, e3 = (np.abs(run(k * base) - ref).max() for k in (1, 2, 4))
    op.close()
    print("errors", e1, e2, e3, "orders", np.log2(e1 / e2), np.log2(e2 / e3))
    assert 3.6 < np.log2(e1 / e2) < 4.4 and 3.5 < np.log2(e2 / e3) < 4.5


@pytest.mark.parametrize("kind,order", [("dry_air", 3), ("argon", 3), ("argon_2T", 2), ("dry_air", 1)])
def test_traces_formed_in_the_flux_epilogue_change_nothing(monkeypatch, kind, order):
    """Stages 2..4 of a step take their face-node traces from the previous stage's k_flux epilogue (RkDev::ta_out) instead
    of a k_traces sweep of their own: the same numbers in the same order, so rk4_step and advance (plain loop and captured
    graph) give bit-identical states with TPSRHS_FUSE_TRACES=0; an external change of x between two steps is seen."""
    import torch
    from tps_amd.rhs_operator import RHSoperator

    if kind == "dry_air":
        c = cases.cyl3d(4, 12, 3, order, capi.NS, capi.VISC_ISOTH)
        U = c.state(seed=5)
        dt = 2e-7
    else:
        c = cases.argon_cyl3d(4, 12, 3, order, kind == "argon_2T", capi.CONSTANT, "arrhenius", capi.VISC_ISOTH)
        U = c.state(seed=5, amp=0.01)
        dt = 2e-9

    def run(fuse):
        monkeypatch.setenv("TPSRHS_FUSE_TRACES", "1" if fuse else "0")
        op = RHSoperator(c.mesh, c.disc, c.physics, c.bcs, stream=torch.cuda.Stream())
        with torch.cuda.stream(op._stream):
            x = torch.tensor(np.ascontiguousarray(U).ravel(), dtype=torch.float64, device=op.device)
            t = 0.0
            for _ in range(2):
                t = op.rk4_step(x, t, dt)
            x[: x.numel() // 7] *= 1.0 + 1e-6  # touched between steps: the next step must sweep its own traces
            t = op.rk4_step(x, t, dt)
            a = x.clone()
            t = op.advance(x, t, dt, 5)[0]  # >= 3 steps on a capturable stream: the captured graph
            torch.cuda.synchronize()
        out = a.cpu().numpy(), x.cpu().numpy()
        op.close()
        return out

    (a1, b1), (a0, b0) = run(True), run(False)
    assert np.array_equal(a1, a0) and np.array_equal(b1, b0)
    assert np.isfinite(b1).all()
