#!/usr/bin/env python3
"""Benchmark of the hot path: RHSoperator::Mult on the 3-D p=3 cylinder workloads.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload argon_p3|cfg2|cfg3|cfg5|torch6]

A "step" is one ``Mult`` (one explicit DG right-hand-side evaluation) over state resident in HBM.
The mesh of the 3-D workloads is the 28x112x16 = 50 176-hex O-grid cylinder of BASELINE.json
configs[1]/[2] per GPU.  Workloads:
  argon_p3 (default)  the metric's "3D p=3 reacting cyl": the reacting argon ternary plasma of configs[2]
                  (ambipolar, single temperature, argon-minimal collision-integral transport with third-order
                  electron conductivity, 2 Arrhenius reactions) at the order and on the mesh of configs[1];
                  6 equations, 3 211 264 nodes
  cfg2            configs[1]: perfect-gas Navier-Stokes, p=3, 5 equations, 3 211 264 nodes
  cfg3            configs[2] itself: the same plasma at p=2
  cfg5            configs[4] on one GPU: axisymmetric 400x500 quads, p=3, two-temperature argon plasma with the
                  argon mixture (collision-integral) transport, reactions and the NEC radiation source, 7 equations;
                  cfg5_const: the same with constant transport (what rounds 1-3 reported as cfg5)
  gll_dry, gll_argon   cfg2 / argon_p3 on the reference's DEFAULT pair (Gauss-Lobatto basis + rules, src/M2ulPhyS.cpp:
                  2671-2672): volume operators through the quadrature points, dense 32 KB inverse mass per hex
  torch6          the mixture of the reference's torch input (plasma.ini): six species, two temperatures, not
                  ambipolar, 11 equations, constant transport as in that input, on the axisymmetric 400x500 mesh of
                  cfg5; torch6_mix: the same mixture with the argon mixture transport
  lte_torch       the table gas of the reference's LTE torch inputs (plasma.lte1d.ini: fluid = lte_table, one-dimensional
                  tables, radiation sink, viscosity-multiplier function), 5 equations, on the same mesh
At N = 1 the JSON line also carries the workloads that were not selected, under `other_workloads`.
  cfg4            configs[3]: perfect-gas Navier-Stokes, p=3, on the 56x224x32 = 401 408-hex cylinder, cut into
                  N spanwise slabs (STRONG scaling: the total work is fixed)
N > 1 (launched by ``torch.distributed.run``, one rank per GPU): every rank owns one block -- spanwise slabs of an
N-times longer cylinder (weak scaling; cfg4: slabs of the same cylinder, strong scaling) -- and exchanges the
traces of its two shared planes with its neighbours through the native RCCL library (libtpsrhs_rccl.so:
ncclSend/ncclRecv groups on the operator's communication stream, no collective on the data path, no Python in the
exchange).  ``--backend gloo --share-gpu`` rehearses the same launches on a one-GPU box through the Python hook
(traces staged via host); the line then says so.  A failing exchange ends the run with a non-zero exit code.

Prints ONE JSON line (rank 0).  ``value`` = DOFs of x processed per second by the whole job / 1e6.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0  # MI355X spec (guides/MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)
HBM_COPY_GBS = 6290.0
# FP64 vector issue: 1024 SIMDs x 2.4 GHz / 4 cycles per wave-instruction (measured issue interval of
# v_fma_f64 / v_mul_f64 / v_add_f64 with two waves per SIMD: tools/microbench/fp64_issue.hip, 3.5-4.2 cycles)
VALU_PEAK_GINST = 1024 * 2.4 / 4.0


def algorithmic_bytes_per_node(neq, dim, p, nc=0):
    """bytes per node per Mult of the two mandatory sweeps (gradient, flux); nc = 1: the
    Gauss-Lobatto rules have one more point per direction."""
    n1 = p + 1
    qf = (((dim - 1) + 2 * p) // 2 + 1 + nc) ** (dim - 1)
    dof = n1**dim
    b_face = 8.0 * 2 * (dim + 1) * qf * dim / dof
    sweep1 = 8.0 * (neq + dim * neq + dim * dim + 1) + 0.5 * b_face
    sweep2 = 8.0 * (2 * neq + dim * neq + dim * dim + 1) + 0.5 * b_face
    return {"k_gradient": sweep1, "k_flux": sweep2, "mult": sweep1 + sweep2}


def workload(name):
    """-> order, physics, bcs(physics), state(X, physics), description, sample-case builder"""
    from tps_amd import capi, cases

    if name in ("cfg2", "cfg4", "gll_dry"):
        return (3, capi.dry_air_physics(capi.NS), lambda ph: cases.cylinder_bcs(capi.VISC_ISOTH, 300.0),
                lambda X, ph: cases.dry_air_state(X, seed=12345),
                "perfect-gas Navier-Stokes (dry air, Sutherland), inlet SUB_DENS_VEL / outlet SUB_P / isothermal "
                "wall (BASELINE.json " + ("configs[1])" if name == "cfg2" else
                                          "configs[1] on the reference's DEFAULT basis / rule pair: Gauss-Lobatto basis, "
                                          "Gauss-Lobatto rules, dense inverse mass)" if name == "gll_dry" else
                                          "configs[3]: 56x224x32 = 401 408 hexes in all, spanwise slabs)"),
                lambda order: cases.cyl3d(7, 28, 4, order, capi.NS, capi.VISC_ISOTH))
    if name in ("argon_p3", "cfg3", "gll_argon"):
        order = 2 if name == "cfg3" else 3
        what = ("BASELINE.json metric: 3D p=3 reacting cylinder = configs[2] physics at configs[1] order"
                if name == "argon_p3" else "the metric's workload on the Gauss-Lobatto basis / rule pair" if name == "gll_argon"
                else "BASELINE.json configs[2]")
        return (order, capi.argon_ternary_physics(capi.NS, False, capi.ARGON_MINIMAL, "arrhenius"),
                lambda ph: cases.plasma_cylinder_bcs(ph, capi.VISC_ISOTH, 3000.0),
                lambda X, ph: cases.plasma_state(X, ph, nvel=3, seed=12345, amp=0.05),
                "reacting argon ternary plasma (Ar+, e, Ar; ambipolar, single temperature), argon-minimal "
                "collision-integral transport with 3rd-order electron conductivity, 2 Arrhenius reactions, inlet "
                f"SUB_DENS_VEL / outlet SUB_P / isothermal wall ({what})",
                lambda order: cases.argon_cyl3d(7, 28, 4, order))
    if name in ("cfg5", "cfg5_const"):
        # configs[4]: "full reacting RHS + transport + radiation source" -- the collision-integral transport of a mixture
        # (GasMixtureTransport, third-order electron conductivity; SURVEY.md 8d: "ConstantTransport -> then
        # GasMixtureTransport"); cfg5_const keeps the constant-transport variant rounds 1-3 reported as cfg5
        tr = capi.CONSTANT if name == "cfg5_const" else capi.ARGON_MIXTURE
        mk = lambda: capi.argon_ternary_physics(capi.NS, True, tr, "tabulated", radiation=True)  # noqa: E731
        return (3, mk(), lambda p: cases.argon_axisym(2, 2, 3, physics=p).bcs,
                lambda X, p: cases.plasma_state(X, p, nvel=3, seed=12345, amp=0.05, vel0=(1.0, 20.0, 3.0)),
                "AXISYMMETRIC (r, z) 400x500 quads, two-temperature argon ternary plasma, "
                + ("constant transport, " if name == "cfg5_const" else "argon mixture transport (collision integrals, 3rd-order electron conductivity), ")
                + "ionisation / three-body recombination on the reference's rate tables (test/inputs/rate-coefficients), "
                "its net-emission table (rad-data/nec_sample.0.h5), inlet / outlet / isothermal wall / axis "
                "(BASELINE.json configs[4] on one GPU)",
                lambda order: cases.argon_axisym(40, 50, order, physics=mk()))
    if name in ("torch6", "torch6_mix"):
        # torch6: the reference's torch input as it is (test/inputs/plasma.ini:163: transport_model = constant);
        # torch6_mix: the same mixture with the argon mixture transport (GasMixtureTransport)
        tr = capi.ARGON_MIXTURE if name == "torch6_mix" else capi.CONSTANT
        mk = lambda: capi.argon_six_species_physics(capi.NS, tr, True, "tabulated", radiation=True)  # noqa: E731
        return (3, mk(), lambda p: cases.argon_axisym(2, 2, 3, physics=p).bcs,
                lambda X, p: cases.plasma_state(X, p, nvel=3, seed=12345, amp=0.05, vel0=(1.0, 20.0, 3.0)),
                "AXISYMMETRIC (r, z) 400x500 quads, the six-species two-temperature argon mixture of the reference's "
                "torch input (test/inputs/plasma.ini: Ar.+1, Ar_m, Ar_r, Ar_p, E, Ar; not ambipolar; 11 equations), "
                + ("argon mixture transport, " if name == "torch6_mix" else "constant transport (plasma.ini:163), ")
                + "the 14 tabulated electron-impact reactions of test/inputs/input.radDecay.ini on the "
                "reference's rate tables, its net-emission table",
                lambda order: cases.argon_axisym(40, 50, order, physics=mk()))
    if name == "lte_torch":
        def lte():
            ph = capi.lte_physics(capi.NS, "rho0p255", radiation=True)
            vs = ph.visc_sponge  # [viscosityMultiplierFunction] of test/inputs/plasma.lte1d.ini:52-58 on this tube
            vs.enabled, vs.width, vs.ratio = 1, 0.02, 20.0
            vs.normal[1], vs.point[1] = 1.0, 0.2
            return ph
        return (3, lte(), lambda p: cases.lte_axisym(2, 2, 3).bcs,
                lambda X, p: cases.lte_state(X, p, seed=12345, amp=0.05),
                "AXISYMMETRIC (r, z) 400x500 quads, the table gas of the reference's LTE torch inputs (fluid = lte_table, "
                "one-dimensional tables: argon thermodynamics of test/inputs/argon_lte_thermo_table.dat at 0.255 kg/m^3, "
                "transport of air_simple_transport_table.dat), net-emission radiation sink, viscosity-multiplier function, "
                "inlet / pressure outlet / isothermal wall / axis",
                lambda order: cases.lte_axisym(40, 50, order, radiation=True))
    raise SystemExit(f"unknown workload {name}")


_LIB_SHA = []


def lib_sha16():
    """first 16 hex digits of the SHA-256 of the kernel library this process loads: ties a counter profile
    (profiles/hbm_traffic.json, tools/profile_summary.py) to the build it was taken from"""
    if not _LIB_SHA:
        import hashlib

        from tps_amd import capi

        # the core library and every kernel-family shared object it can load (its own directory, and the
        # directories of TPSRHS_FAMILY_PATH, which come first)
        import glob

        libs = [capi.LIB_PATH]
        for d in [p for p in os.environ.get("TPSRHS_FAMILY_PATH", "").split(":") if p] + [os.path.dirname(capi.LIB_PATH)]:
            libs += sorted(glob.glob(os.path.join(d, "libtpsrhs_plasma_*.so")))
        h = hashlib.sha256()
        for lib in libs:
            with open(lib, "rb") as f:
                for chunk in iter(lambda: f.read(1 << 22), b""):
                    h.update(chunk)
        _LIB_SHA.append(h.hexdigest()[:16])
    return _LIB_SHA[0]


def cpu_baseline(neq, order, sample_case, budget_s=10.0):
    """The oracle (CPU restatement, reference-faithful dense formulation: kind "port") timed on this host on
    a bounded sample of the same workload -- a 7x28x4 = 784-element O-grid block of the cylinder at the same
    order, physics and boundary conditions -- on all cores of the GPU's share of the host and on one thread
    (one MPI rank of the reference), SURVEY 8(d).  About `budget_s` seconds each."""
    from oracle_lib import Oracle

    c = sample_case(order)
    U = c.state()
    ndofs = U.shape[1]

    def timed(threads):
        o = Oracle(c.mesh, c.disc, c.physics, c.bcs, threads=threads)
        o.mult(U)  # warm-up
        t0 = time.perf_counter()
        n = 0
        while n < 3 or (time.perf_counter() - t0 < budget_s and n < 2000):
            o.mult(U)
            n += 1
        dt = (time.perf_counter() - t0) / n
        return {"value": ndofs * neq / dt / 1e6, "unit": "MDOF/s", "cores": threads, "mult_calls": n,
                "evals_per_s_on_sample": 1.0 / dt}

    # the GPU box gives one GPU's share of the host (16 cores); never oversubscribe
    threads = min(len(os.sched_getaffinity(0)), 16)
    allc, one = timed(threads), timed(1)
    return {"value": allc["value"], "unit": "MDOF/s", "cores": threads, "kind": "port",
            "sample": f"cyl3d O-grid 7x28x4 = {c.mesh.num_elements} hexes, p={order}, {c.description}, "
                      f"{ndofs} nodes, {allc['mult_calls']} Mult calls on {threads} threads (OpenMP over "
                      f"elements/faces/nodes) and {one['mult_calls']} on one thread",
            "evals_per_s_on_sample": allc["evals_per_s_on_sample"],
            "single_thread": {"value": one["value"], "unit": "MDOF/s", "cores": 1,
                              "evals_per_s_on_sample": one["evals_per_s_on_sample"]}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--nr", type=int, default=28)
    ap.add_argument("--ntheta", type=int, default=112)
    ap.add_argument("--nz", type=int, default=16)
    ap.add_argument("--order", type=int, default=0, help="override the workload's polynomial order")
    ap.add_argument("--workload", default="argon_p3", choices=["argon_p3", "cfg2", "cfg3", "cfg4", "cfg5", "cfg5_const", "torch6", "torch6_mix", "gll_dry", "gll_argon", "lte_torch"])
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + --share-gpu rehearses the multi-rank path on a one-GPU box (traces staged via host)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal only: every rank uses device 0")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-workloads", action="store_true",
                    help="N=1 only: skip the secondary workloads reported under other_workloads")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    from tps_amd import capi, cases, meshgen
    from tps_amd.halo import HaloExchange
    from tps_amd.rhs_operator import RHSoperator, node_coordinates

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; there is no CPU path")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    halo = None
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            from tps_amd.halo_rccl import RcclHalo

            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            halo = RcclHalo(local_rank)  # C function pointers into libtpsrhs_rccl.so; no fallback
        else:
            dist.init_process_group("gloo")
            halo = HaloExchange(device=torch.device("cuda", local_rank))

    def barrier():
        if world > 1:
            dist.barrier()

    def run(wname, steps, warmup):
        """-> the JSON fields of one workload (rank 0; None elsewhere)"""
        order, physics, make_bcs, make_state, description, sample_case = workload(wname)
        order = args.order or order
        axisym = wname in ("cfg5", "cfg5_const", "torch6", "torch6_mix", "lte_torch")
        strong = wname == "cfg4"
        if axisym:  # (r, z) tube 0.05 x 0.25 per rank, 400 x 500 quads; axial slabs at N > 1 (weak scaling)
            mesh = meshgen.annulus_quad_slab(400, 500, rank, world, r_in=0.0, r_out=0.05, length_local=0.25)
        elif strong:  # the 56x224x32 cylinder of configs[3] in `world` spanwise slabs
            if 32 % world:
                raise SystemExit("cfg4: the 32 spanwise layers must divide by the number of ranks")
            mesh = meshgen.ogrid_cylinder_slab(56, 224, 32 // world, rank, world, span_local=4.0 / world)
        else:
            mesh = meshgen.ogrid_cylinder_slab(args.nr, args.ntheta, args.nz, rank, world)
        gll = wname.startswith("gll_")
        disc = capi.Disc(order, 1 if gll else 0, 1 if gll else 0, 1 if axisym else 0, 0)
        bcs = make_bcs(physics)
        X = node_coordinates(mesh, order, 1 if gll else 0)
        U = make_state(X, physics)
        del X
        op = RHSoperator(mesh, disc, physics, bcs, device=local_rank, halo=halo)
        neq, ndofs = op.num_equation, op.NDofs
        x = torch.tensor(U.ravel(), dtype=torch.float64, device=op.device)
        y = torch.empty_like(x)
        del U
        for _ in range(warmup):
            op.Mult(x, y)
        torch.cuda.synchronize()
        barrier()
        op.enable_kernel_timing(True)  # hipEvent records on the operator's stream, no synchronisation
        torch.cuda.synchronize()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            op.Mult(x, y)
        torch.cuda.synchronize()
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=op.device if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        ktimes = op.kernel_times()  # ms, averaged over the timed Mults
        mult_ms = sorted(op.mult_times())  # device time of each timed Mult (the last 128 at most)
        op.enable_kernel_timing(False)
        finite = bool(torch.isfinite(y).all().item())
        comm_exposed_ms = None
        if world > 1 and halo is not None:
            # exposed communication: the same partitioned launches with the exchange switched off
            # (stale halo traces: timing only), max over ranks like the main number
            halo.skip = True
            for _ in range(2):
                op.Mult(x, y)
            torch.cuda.synchronize()
            barrier()
            t1 = time.perf_counter()
            for _ in range(steps):
                op.Mult(x, y)
            torch.cuda.synchronize()
            barrier()
            dt_nc = time.perf_counter() - t1
            halo.skip = False
            t = torch.tensor([dt_nc], dtype=torch.float64, device=op.device if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            comm_exposed_ms = 1e3 * (dt - float(t.item())) / steps
        rk4 = None
        if world == 1 and wname == args.workload:
            # next row of the scope table: the RK4 time loop on the device (tpsrhs_advance),
            # a constant, tiny dt so that the state stays where the Mult timing left it
            nst = max(2, min(10, steps // 4))
            op.advance(x, 0.0, 1.0e-10, 1, True)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            _, _, bad = op.advance(x, 0.0, 1.0e-10, nst, True)
            t_loop = time.perf_counter() - t1
            rk4 = {"rk4_steps_per_s": nst / t_loop, "ms_per_rk4_step": 1e3 * t_loop / nst, "steps": nst,
                   "nan_entries": int(bad), "mult_per_step": 4}
        comm = None
        if world > 1:
            if hasattr(halo, "stats"):
                st = halo.stats()
                calls, sent, peers = st["halo_calls"], st["bytes_sent"], st["peers_seen"]
                nranks, reduce_comm = st["nranks"], st["reduce_comm"]  # ncclCommCount of the exchange communicator
            else:
                calls, sent, peers = halo.calls, halo.bytes_sent, len({(rank - 1) % world, (rank + 1) % world})
                nranks, reduce_comm = dist.get_world_size(), "torch.distributed (gloo)"
            comm = {"backend_used": halo.backend if hasattr(halo, "stats") else f"{halo.backend} (Python hook, staged through host)",
                    "nranks": nranks, "ranks_seen": peers, "reduce_comm": reduce_comm, "halo_calls": calls,
                    "halo_bytes_per_mult": 2.0 * sent / max(calls, 1)}  # two exchanges (TA, TB) per Mult
        op.close()
        del x, y
        if rank != 0:
            return None
        ms_per_step = 1e3 * dt / steps
        evals_per_s = steps / dt
        value = world * ndofs * neq * evals_per_s / 1e6
        alg = algorithmic_bytes_per_node(neq, mesh.dim, order, 1 if gll else 0)
        if gll:  # the dense NPE x NPE inverse mass of every element is streamed by both sweeps
            npe = (order + 1) ** mesh.dim
            for k in ("k_gradient", "k_flux"):
                alg[k] += 8.0 * npe
            alg["mult"] += 16.0 * npe
        dom = max((k for k in ktimes if k in alg), key=lambda k: ktimes[k])
        achieved = alg[dom] * ndofs / (ktimes[dom] * 1e-3) / 1e9
        traffic = valu_insts = None
        counters_from = "no counter profile of this workload in profiles/hbm_traffic.json"
        try:  # written by tools/profile_summary.py from the rocprofv3 --pmc passes of this workload
            tj = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json"))).get(wname, {})
            if tj and tj.get("lib_sha16") != lib_sha16():
                # the counters describe another build of the kernels: report none rather than stale ones
                counters_from = f"profiles/hbm_traffic.json holds counters of build {tj.get('lib_sha16')}, this is {lib_sha16()}"
            elif tj.get("nodes") == ndofs:
                traffic = tj.get("bytes_per_launch", {}).get(dom)
                valu_insts = tj.get("valu_insts_per_launch", {}).get(dom)
                counters_from = f"profiles/hbm_traffic.json, build {tj.get('lib_sha16')} (the library loaded here)"
        except Exception as exc:
            counters_from = f"profiles/hbm_traffic.json unreadable: {exc}"
        # second roofline of a kernel that is not bandwidth-bound: FP64 vector issue.  achieved = VALU
        # wave-instructions per launch (SQ_INSTS_VALU of the committed PMC pass) / the live kernel time.
        valu = None
        if valu_insts:
            ginst = valu_insts / (ktimes[dom] * 1e-3) / 1e9
            valu = {"bound": "fp64-valu-issue", "kernel": dom, "achieved": ginst, "peak": VALU_PEAK_GINST,
                    "unit": "G wave-instructions/s", "frac": ginst / VALU_PEAK_GINST,
                    "valu_insts_per_launch": valu_insts}
        else:
            valu = {"bound": "fp64-valu-issue", "kernel": dom, "achieved": None, "peak": VALU_PEAK_GINST,
                    "unit": "G wave-instructions/s", "frac": None, "valu_insts_per_launch": None}
        valu["counters_from"] = counters_from
        res = {
            "value": value, "ms_per_step": ms_per_step, "steps": steps, "warmup": warmup,
            "config": {"workload": (f"{wname}: " + ("" if axisym else
                                                    (f"cyl3d O-grid 56x224x{32 // world} hexes per GPU, " if strong else
                                                     f"cyl3d O-grid {args.nr}x{args.ntheta}x{args.nz} hexes per GPU, ")) +
                                    f"p={order}, " + ("Gauss-Lobatto basis + rules" if gll else "GL basis + GL rule") + f", {description}"),
                       "elements_per_gpu": mesh.num_elements, "nodes_per_gpu": ndofs, "num_equation": neq,
                       "partition": (f"{world} spanwise slabs; face traces of the shared planes exchanged by "
                                     f"{comm['backend_used']}") if world > 1 else "single GPU"},
            "scaling": "strong" if strong else "weak", "comm": comm,
            "rhs_evals_per_s": evals_per_s, "mnodes_per_s": world * ndofs * evals_per_s / 1e6, "kernel_ms": ktimes,
            "ms_per_mult_median_events": mult_ms[len(mult_ms) // 2] if mult_ms else None,
            "finite": finite, "time_loop": rk4, "comm_exposed_ms": comm_exposed_ms, "roofline_valu": valu,
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "frac_of_measured_copy": achieved / HBM_COPY_GBS,
                         "traffic": traffic, "algorithmic_bytes_per_node": alg[dom],
                         "mult_algorithmic_GBps": alg["mult"] * ndofs / (ms_per_step * 1e-3) / 1e9},
        }
        return res, (neq, order, sample_case)

    r = run(args.workload, args.steps, args.warmup)
    others = {}
    if world == 1 and not args.no_other_workloads:
        for wname in ("argon_p3", "cfg2", "cfg3", "cfg5", "cfg5_const", "torch6", "torch6_mix", "gll_dry", "lte_torch"):
            if wname != args.workload:
                # the same K timed steps as the headline and at least 10 warm-ups: comparable round to round and
                # with the profiles/ of the same command
                o, _ = run(wname, args.steps, max(args.warmup, 10))
                others[wname] = {k: o[k] for k in ("value", "ms_per_step", "steps", "warmup", "rhs_evals_per_s", "kernel_ms",
                                                  "ms_per_mult_median_events", "finite", "roofline", "roofline_valu")}
                others[wname]["unit"] = "MDOF/s"
                others[wname]["workload"] = o["config"]["workload"]
    if rank == 0:
        res, (neq, order, sample_case) = r
        out = {
            "metric": "DG RHS evals/sec (MDOF/s) for 3D p=3 reacting cyl at 1/2/4/8 MI355X",
            "value": res["value"], "unit": "MDOF/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": res["scaling"], "vs_baseline": None,
            "dtype": "f64", "data": "synthetic", "config": res["config"],
            "rhs_evals_per_s": res["rhs_evals_per_s"], "mnodes_per_s": res["mnodes_per_s"], "kernel_ms": res["kernel_ms"],
            "finite": res["finite"],
            "ms_per_mult_median_events": res["ms_per_mult_median_events"],
            "roofline": res["roofline"],
        }
        out["roofline_valu"] = res["roofline_valu"]
        out["lib_sha16"] = lib_sha16()
        if res.get("time_loop"):
            out["time_loop"] = res["time_loop"]
        if res.get("comm"):
            out.update(res["comm"])  # backend_used, ranks_seen, halo_calls, halo_bytes_per_mult
        if res.get("comm_exposed_ms") is not None:
            out["comm_exposed_ms"] = res["comm_exposed_ms"]  # ms_per_step minus the same launches without the exchange
        if others:
            out["other_workloads"] = others
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(neq, order, sample_case)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
