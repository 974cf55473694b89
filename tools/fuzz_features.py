"""Randomised sweep over the widening rows (run on a GPU box): dry air or argon, random order / mesh, random
subset of {non-reflecting inlet/outlet, forcing terms, Joule heating, Roe flux, device time loop}; HIP vs oracle
over several consecutive Mult calls (or an advance of a few steps).  Not part of the test suite.
    python tools/fuzz_features.py [ncases] [seed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle_lib import Oracle  # noqa: E402
from tps_amd import capi, cases, meshgen  # noqa: E402
from tps_amd.rhs_operator import RHSoperator, node_coordinates  # noqa: E402

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
TOL = 2e-11
worst, t0 = 0.0, time.time()


def nr_bc(attr, cat, typ, data, tangent, area=0.0):
    return capi.make_bc(attr, cat, typ, list(data) + [0.0] * (4 - len(data)) + list(tangent) + [area])


def target(rho, vel, p):
    return [rho] + [rho * v for v in vel] + [p / 0.4 + 0.5 * rho * sum(v * v for v in vel)]


for it in range(ncases):
    geo = rng.choice(["cyl3d", "chan2d", "plasma3d", "axi"], p=[0.4, 0.3, 0.1, 0.2])
    order = int(rng.integers(1, 6 if geo in ("cyl3d", "chan2d") else (4 if geo == "axi" else 3)))
    seed = int(rng.integers(1, 1000))
    feats = []
    try:
        if geo == "cyl3d":
            c = cases.cyl3d(int(rng.integers(3, 6)), int(rng.integers(8, 13)), int(rng.integers(3, 5)), order, capi.NS,
                            int(rng.choice([capi.INV, capi.VISC_ADIAB, capi.VISC_ISOTH])))
            c.physics.dry_air.visc_mult = float(rng.choice([1.0, 500.0]))
            mesh, disc, ph, bcs = meshgen.scramble_orientations(c.mesh, seed), c.disc, c.physics, c.bcs
            U = cases.dry_air_state(node_coordinates(mesh, order), seed=seed)
            tang, dim = (0.0, 0.0, 1.0), 3
        elif geo == "chan2d":
            attrs = {(0, 0): 1, (0, 1): 2, (1, 0): 3, (1, 1): 3}
            mesh = meshgen.scramble_orientations(meshgen.box_quad(int(rng.integers(3, 8)), int(rng.integers(3, 7)), lengths=(1.0, 0.7),
                                                                  periodic=(False, False), bdr_attr=attrs, warp=0.05), seed)
            disc = capi.Disc(order, 0, 0, 0, 0)
            ph = capi.dry_air_physics(capi.NS if rng.random() < 0.8 else capi.EULER, visc_mult=300.0)
            bcs = [capi.make_bc(1, capi.INLET, capi.SUB_DENS_VEL, [1.2, 20.0, 0.0, 0.0]), capi.make_bc(2, capi.OUTLET, capi.SUB_P, [101300.0]),
                   capi.make_bc(3, capi.WALL, int(rng.choice([capi.INV, capi.VISC_ADIAB, capi.VISC_ISOTH])), [300.0])]
            U = cases.dry_air_state(node_coordinates(mesh, order), seed=seed)
            tang, dim = (0.0, 1.0, 0.0), 2
            if rng.random() < 0.3:
                disc.use_roe = 1
                feats.append("roe")
        elif geo == "axi":  # axisymmetric tube: dry air or an argon mixture, with the mixing-length model
            wall = int(rng.choice([capi.INV, capi.VISC_ADIAB, capi.VISC_ISOTH]))
            fl = rng.choice(["dry", "ternary", "six", "lte"])
            if fl == "lte":  # the table gas (fluid = lte_table, one-dimensional tables)
                if wall == capi.VISC_ADIAB and rng.random() < 0.5:
                    wall = capi.VISC_ISOTH
                c = cases.lte_axisym(int(rng.integers(3, 7)), int(rng.integers(3, 8)), order, capi.NS, wall,
                                     radiation=bool(rng.random() < 0.5), density=str(rng.choice(["rho0p005", "rho0p255"])))
                c.disc.use_bc_in_grad = int(rng.random() < 0.5)
                amp = 0.05
            elif fl == "dry":
                c = cases.dry_air_axisym(int(rng.integers(3, 7)), int(rng.integers(3, 8)), order, capi.NS, wall)
                c.physics.dry_air.visc_mult = float(rng.choice([1.0, 50.0]))
                amp = 0.05
            else:
                ph_ = (capi.argon_ternary_physics(capi.NS, bool(rng.random() < 0.5), int(rng.choice([capi.CONSTANT, capi.ARGON_MINIMAL])), "arrhenius")
                       if fl == "ternary" else capi.argon_six_species_physics(capi.NS, int(rng.choice([capi.CONSTANT, capi.ARGON_MIXTURE])), True, True))
                c = cases.argon_axisym(int(rng.integers(3, 7)), int(rng.integers(3, 8)), order, physics=ph_, wall_type=wall)
                amp = 0.005 if order == 1 else 0.01
            mesh, disc, ph, bcs = c.mesh, c.disc, c.physics, c.bcs
            U = c.state(seed=seed, amp=amp)
            tang, dim = None, 2
            feats.append(fl)
            if rng.random() < 0.5:  # [viscosityMultiplierFunction] in the 2-D heavy kernels
                vs = ph.visc_sponge
                vs.enabled, vs.width, vs.ratio = 1, float(rng.uniform(0.01, 0.08)), float(rng.uniform(2.0, 30.0))
                vs.normal[0], vs.normal[1] = float(rng.uniform(-1, 1)), float(rng.uniform(0.2, 1.0))
                vs.point[0], vs.point[1] = float(rng.uniform(0.0, 0.05)), float(rng.uniform(0.05, 0.2))
                feats.append("sponge")
        else:
            two_t = bool(rng.random() < 0.5)
            c = cases.argon_cyl3d(4, int(rng.integers(8, 12)), 3, order, two_t, int(rng.choice([capi.CONSTANT, capi.ARGON_MINIMAL])), "arrhenius",
                                  capi.VISC_ISOTH)
            mesh, disc, ph, bcs = c.mesh, c.disc, c.physics, c.bcs
            U = c.state(seed=seed, amp=0.005)
            tang, dim = None, 3
        dry = geo in ("cyl3d", "chan2d")
        if dry and rng.random() < 0.6:
            typ = int(rng.choice([capi.SUB_P_NR, capi.SUB_MF_NR, capi.SUB_MF_NR_PW]))
            bcs[1] = nr_bc(2, capi.OUTLET, typ, [101000.0] if typ == capi.SUB_P_NR else [24.0 * 3.0], tang, 3.0)
            feats.append(f"out{typ}")
            disc.ref_length = float(rng.choice([0.5, 2.0]))
        if dry and rng.random() < 0.4:
            typ = int(rng.choice([capi.SUB_DENS_VEL_NR, capi.SUB_VEL_CONST_ENT]))
            bcs[0] = nr_bc(1, capi.INLET, typ, [1.21, 19.0, 0.4, -0.2 if dim == 3 else 0.0], tang)
            feats.append(f"in{typ}")
        forcing = None
        if rng.random() < 0.6:
            kw = {}
            if rng.random() < 0.5:
                kw["pressure_gradient"] = tuple(rng.uniform(-5, 5, 3))
            if rng.random() < 0.5:
                kw["heat_sources"] = [dict(value=float(rng.uniform(-1e4, 1e5)), radius=float(rng.uniform(0.2, 2.0)),
                                           point1=tuple(rng.uniform(-1, 1, 3)), point2=tuple(rng.uniform(1.5, 3, 3)))]
            if rng.random() < 0.5:
                tu = target(1.15, (18.0, 1.0, -0.5)[:dim], 1.0e5) if dry else list(U[:, 3])  # (table gas, mixtures: a state of the field)
                x0 = 5.0 if geo != "chan2d" else 0.6
                kw["sponge_zones"] = [dict(type=capi.SPONGE_PLANAR, normal=(-1.0, float(rng.uniform(-0.2, 0.2)), 0.0),
                                           point0=(2 * x0, 0.0, 0.0), point_init=(x0, 0.0, 0.0), mult_factor=float(rng.uniform(0.2, 2.0)),
                                           target_U=tu)]
                if dry and rng.random() < 0.4:  # mixed-out target: the plane through point_init, a generous node tolerance
                    kw["sponge_zones"][0].update(solution_type=capi.SPONGE_MIXEDOUT, tol=1.0 if geo == "cyl3d" else 0.12)
                    feats.append("mixedout")
            if rng.random() < 0.4:  # PassiveScalar: the last equation, inside a ball
                ctr = (5.0, 0.0, 0.0) if geo in ("cyl3d", "plasma3d") else ((0.02, 0.1, 0.0) if geo == "axi" else (0.5, 0.3, 0.0))
                kw["passive_scalars"] = [dict(xyz=ctr, radius=float(rng.uniform(0.5, 6.0) if dim == 3 else rng.uniform(0.03, 0.4)),
                                              value=float(rng.uniform(-2, 300)))]
            forcing = capi.make_forcing(**kw)
            feats.append("forcing:" + ",".join(sorted(k[:4] for k in kw)))
        joule = rng.uniform(-1e4, 5e4, U.shape[1]) if (dim == 3 and rng.random() < 0.3) else None
        if joule is not None:
            feats.append("joule")
        use_advance = dry and rng.random() < 0.3  # (an explicit step at this dt is unstable for the stiff plasma cases)
        dt = float(rng.choice([1e-5, 1e-4])) if not use_advance else 2e-6
        desc = f"{geo} p={order} seed={seed} [{' '.join(feats)}]" + (" advance" if use_advance else "")
        o = Oracle(mesh, disc, ph, bcs)
        op = RHSoperator(mesh, disc, ph, bcs)
        for obj, setf, setj in ((o, o.set_forcing, o.set_joule_heating), (op, op.setForcing, None)):
            setf(forcing)
        o.set_joule_heating(joule)
        jt = None if joule is None else torch.tensor(joule, dtype=torch.float64, device=op.device)
        op.setJouleHeating(jt)
        if geo == "axi" and rng.random() < 0.8:
            dist = np.ascontiguousarray(0.05 - node_coordinates(mesh, order)[0])  # distance to the tube wall
            prm = dict(max_mixing_length=float(rng.choice([1e-3, 4e-3, 1.0])), pr_ratio=float(rng.uniform(0.5, 1.2)),
                       bulk_multiplier=float(rng.choice([0.0, 1.0])))
            o.set_mixing_length(dist, **prm)
            dt_ = torch.tensor(dist, dtype=torch.float64, device=op.device)
            op.setMixingLength(dt_, **prm)
            desc += " mixlen"
        x = torch.tensor(np.ascontiguousarray(U).ravel(), dtype=torch.float64, device=op.device)
        errs = []
        if use_advance:
            var = bool(rng.random() < 0.5)
            ref, tr_, dtr, _ = o.advance(U, 0.0, dt, 2, not var, 0.05, 0.02)
            tg, dtg, _ = op.advance(x, 0.0, dt, 2, not var, 0.05, 0.02)
            got = x.cpu().numpy().reshape(U.shape)
            errs.append((np.abs(got - ref).max(axis=1) / np.abs(ref).max(axis=1)).max() * 100)  # states: 1e-13 scale
            errs.append(abs(dtg - dtr) / dtr + abs(tg - tr_) / max(tr_, 1e-300))
        else:
            o.set_dt(dt)
            op.setDt(dt)
            y = torch.empty_like(x)
            for call in range(3):
                ref = o.mult(U)
                if not np.isfinite(ref).all():
                    raise FloatingPointError("oracle not finite (inadmissible input)")
                op.Mult(x, y)
                got = y.cpu().numpy().reshape(U.shape)
                scale = np.abs(ref).max(axis=1)
                scale[1:1 + dim] = scale[1:1 + dim].max()
                errs.append((np.abs(got - ref).max(axis=1) / np.maximum(scale, 1e-300)).max())
        op.close()
        err = max(errs)
        tol = TOL * (10 if (geo == "plasma3d" or (geo == "axi" and feats[0] != "dry")) else 1)
        worst = max(worst, err / tol)
        print(f"[{it}] {desc}: {err:.2e}" + ("" if err < tol else "   <<<<<< FAIL"), flush=True)
    except Exception as exc:  # noqa: BLE001
        print(f"[{it}] {geo} p={order} seed={seed} {feats}: EXCEPTION {type(exc).__name__}: {exc}", flush=True)
print(f"worst err/tol = {worst:.3f}; {time.time() - t0:.0f} s")
