"""Partitioned Mult on the GPU: 2-3 ranks (sharing the one GPU of the test box, traces staged over
gloo) must reproduce the serial oracle on the unpartitioned mesh.  Exercises the interior/halo block
split, the second (communication) stream and the canonical-frame packing of the shared faces."""
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from parity_util import RHS_RTOL, oracle_mult, rel_maxnorm
from tps_amd import capi, cases, meshgen
from tps_amd.rhs_operator import node_coordinates

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


MIXEDOUT = dict(type=capi.SPONGE_PLANAR, solution_type=capi.SPONGE_MIXEDOUT, normal=(-1.0, 0.0, 0.0), point0=(1.9, 0.0, 0.0),
                point_init=(1.0, 0.0, 0.0), tol=0.06, mult_factor=2.0, target_U=[])


def _case(world, kind):
    if kind == "mixedout":  # mixed-out sponge zone: the plane sums are added over the ranks (tpsrhs_reduce_fn)
        full = meshgen.scramble_orientations(meshgen.box_hex(8, 3, 3, lengths=(2.0, 1.0, 0.5)), 3)
        owner = (np.arange(full.num_elements) * 5 // 3) % world  # the plane nodes are spread over every rank
        ph = capi.dry_air_physics(capi.NS)
        Ug = cases.dry_air_state(node_coordinates(full, 2), seed=4, amp=0.1)
        return full, owner, 2, ph, [], Ug
    if kind == "slab_thin":  # the shape of cfg4 at N = 8 (a few layers per rank: EVERY element touches a shared plane)
        full = meshgen.ogrid_cylinder(4, 12, 2 * world, span=2.0 * world)
        order = 3
        ph = capi.dry_air_physics(capi.NS, visc_mult=2000.0)
        bcs = cases.cylinder_bcs(capi.VISC_ISOTH)
        Ug = cases.dry_air_state(node_coordinates(full, order), seed=5)
        return full, "slab2", order, ph, bcs, Ug
    if kind == "slab":  # the weak-scaling partition of bench.py: spanwise slabs, both neighbours may be one rank
        full = meshgen.ogrid_cylinder(4, 12, 3 * world, span=2.0 * world)
        order = 3
        ph = capi.dry_air_physics(capi.NS, visc_mult=2000.0)
        bcs = cases.cylinder_bcs(capi.VISC_ISOTH)
        Ug = cases.dry_air_state(node_coordinates(full, order), seed=5)
        return full, "slab", order, ph, bcs, Ug
    if kind == "axisym_slab":  # bench.py's partition of the axisymmetric workloads: axial slabs of the (r, z) block
        ph = capi.argon_ternary_physics(capi.NS, True, capi.CONSTANT, "arrhenius", radiation=True)
        full = meshgen.annulus_quad(6, 3 * world, r_out=0.05, length=0.08 * world)
        c = cases.argon_axisym(6, 3, 2, physics=ph)  # boundary conditions of the tube
        Ug = cases.plasma_state(node_coordinates(full, 2), ph, nvel=3, seed=8, amp=0.01, vel0=(1.0, 20.0, 3.0))
        return full, "axislab", 2, ph, c.bcs, Ug
    if kind == "axisym_2T":  # 2-D shared edges, axisymmetric two-temperature plasma, three contiguous parts
        c = cases.argon_axisym(6, 9, 2, True, capi.CONSTANT, "arrhenius", True, capi.VISC_ISOTH, r_in=0.0)
        full = meshgen.scramble_orientations(c.mesh, 5)
        Ug = cases.plasma_state(node_coordinates(full, 2), c.physics, nvel=3, seed=6, amp=0.01, vel0=(1.0, 20.0, 3.0))
        return full, None, 2, c.physics, c.bcs, Ug
    full = meshgen.scramble_orientations(meshgen.ogrid_cylinder(4, 12, 4), 21)
    if kind in ("dry_air", "dry_air_nr"):
        owner = (np.arange(full.num_elements) * 7 // 5) % world  # irregular partition: every block is a halo block
        order = 3
        ph = capi.dry_air_physics(capi.NS, visc_mult=2000.0)
        bcs = cases.cylinder_bcs(capi.VISC_ISOTH)
        if kind == "dry_air_nr":  # non-reflecting outlet: patch mean summed over the ranks (tpsrhs_reduce_fn)
            bcs[1] = capi.make_bc(2, capi.OUTLET, capi.SUB_P_NR, [101000.0, 0, 0, 0, 0.0, 0.0, 1.0, 0.0])
        Ug = cases.dry_air_state(node_coordinates(full, order), seed=5)
    else:
        owner = None  # contiguous thirds: interior and halo blocks, 2 elements per workgroup at p = 2
        order = 2
        ph = capi.argon_ternary_physics(capi.NS, True, capi.CONSTANT, "arrhenius")
        bcs = cases.plasma_cylinder_bcs(ph, capi.VISC_ISOTH)
        Ug = cases.plasma_state(node_coordinates(full, order), ph, nvel=3, seed=5, amp=0.01)
    return full, owner, order, ph, bcs, Ug


def _worker(rank, world, port, q, kind, backend="gloo"):
    try:
        import torch
        import torch.distributed as dist

        from tps_amd.halo import HaloExchange
        from tps_amd.rhs_operator import RHSoperator

        dev = rank if backend == "nccl" else 0  # nccl (= RCCL): one GPU per rank; gloo: the ranks share GPU 0
        torch.cuda.set_device(dev)
        if backend == "nccl":
            dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
        full, owner, order, ph, bcs, Ug = _case(world, kind)
        if isinstance(owner, str):
            if owner == "axislab":
                part = meshgen.annulus_quad_slab(6, 3, rank, world, r_out=0.05, length_local=0.08)
            elif owner == "slab2":
                part = meshgen.ogrid_cylinder_slab(4, 12, 2, rank, world)
            else:
                part = meshgen.ogrid_cylinder_slab(4, 12, 3, rank, world)
            gel = np.arange(part.num_elements) + rank * part.num_elements
        else:
            part = meshgen.partition(full, world, owner)[rank]
            gel = part.global_elements
        disc = capi.Disc(order, 0, 0, 1 if kind.startswith("axisym") else 0, 0)
        npe = (order + 1) ** full.dim
        idx = (gel[:, None] * npe + np.arange(npe)[None, :]).ravel()
        U = Ug[:, idx]  # the rank's rows of ONE global field
        if backend == "nccl":  # the native library: C function pointers, ncclSend / ncclRecv groups
            from tps_amd.halo_rccl import RcclHalo

            halo = RcclHalo(dev)
        else:
            halo = HaloExchange(device=torch.device("cuda", dev))
        op = RHSoperator(part, disc, ph, bcs, device=dev, halo=halo)
        if kind == "mixedout":
            op.setForcing(capi.make_forcing(sponge_zones=[MIXEDOUT]))
        x = torch.tensor(np.ascontiguousarray(U).ravel(), dtype=torch.float64, device=op.device)
        y = torch.empty_like(x)
        if kind.endswith("_nr"):  # second call: the boundary state of the first one is in use
            op.setDt(NR_DT)
            op.Mult(x, y)
        op.Mult(x, y, want_max_char_speed=True)
        torch.cuda.synchronize()
        out = y.cpu().numpy().reshape(U.shape)
        g = op.getGradients().cpu().numpy()
        if kind == "dry_air_nr":  # ... and the device time loop with a CFL-controlled dt: MIN over the ranks
            t_end, dt_next, bad = op.advance(x, 0.0, ADV[0], ADV[1], False, ADV[2], ADV[3])
            adv = (x.cpu().numpy().reshape(U.shape), t_end, dt_next, bad)
        else:
            adv = None
        dist.barrier()
        op.close()
        dist.destroy_process_group()
        q.put((rank, "ok", idx, out, g, op.max_char_speed, adv))
    except Exception:  # pragma: no cover
        import traceback

        q.put((rank, traceback.format_exc(), None, None, None, None, None))


NR_DT = 3.0e-4
ADV = (2.0e-5, 3, 0.1, 0.05)  # dt0, steps, CFL, hmin of the advance() leg


@pytest.mark.parametrize("world,kind", [(2, "dry_air"), (3, "argon_2T"), (2, "slab"), (4, "slab"), (3, "axisym_2T"),
                                        (3, "dry_air_nr"), (3, "axisym_slab"), (3, "mixedout"), (5, "slab_thin")])  # 5 ranks + the test process = the 6 processes a GPU box allows
def test_ranks_match_serial_oracle(world, kind):
    _run_ranks(world, kind, "gloo")


@pytest.mark.parametrize("world,kind", [(2, "slab"), (2, "dry_air_nr")])
def test_ranks_match_serial_oracle_over_rccl(world, kind):
    """The same on one GPU per rank through libtpsrhs_rccl.so (ncclSend / ncclRecv groups of the traces straight
    from device memory, ncclAllReduce of the boundary means and of dt): needs as many GPUs as ranks, so it is
    skipped on the one-GPU test boxes and runs wherever a multi-GPU node executes the suite."""
    import torch

    if torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs")
    _run_ranks(world, kind, "nccl")


def _run_ranks(world, kind, backend):
    full, owner, order, ph, bcs, Ug = _case(world, kind)
    if kind.endswith("_nr"):
        from oracle_lib import Oracle

        o = Oracle(full, capi.Disc(order, 0, 0, 0, 0), ph, bcs)
        o.set_dt(NR_DT)
        o.mult(Ug)
        ref = {"y": o.mult(Ug), "gradUp": o.gradients(), "max_char_speed": o.max_char_speed}
        ref_adv = o.advance(Ug, 0.0, ADV[0], ADV[1], False, ADV[2], ADV[3])
    elif kind == "mixedout":
        from oracle_lib import Oracle

        o = Oracle(full, capi.Disc(order, 0, 0, 0, 0), ph, bcs)
        y_plain = o.mult(Ug)
        o.set_forcing(capi.make_forcing(sponge_zones=[MIXEDOUT]))
        ref = {"y": o.mult(Ug), "gradUp": o.gradients(), "max_char_speed": o.max_char_speed}
        assert np.abs(ref["y"] - y_plain).max(axis=1).min() > 0.0  # the zone is active in every equation
    else:
        ref = oracle_mult(full, capi.Disc(order, 0, 0, 1 if kind.startswith("axisym") else 0, 0), ph, bcs, Ug)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, kind, backend)) for r in range(world)]
    for p in procs:
        p.start()
    res = []
    try:  # a rank that fails leaves its peers blocked in the exchange: collect what arrives, then end them all
        for _ in procs:
            res.append(q.get(timeout=300))
            if res[-1][1] != "ok":
                break
    finally:
        for p in procs:
            p.join(timeout=60 if len(res) == len(procs) and all(r[1] == "ok" for r in res) else 5)
        for p in procs:
            if p.is_alive():
                p.terminate()
                p.join(timeout=10)
            if p.is_alive():
                p.kill()
                p.join()
    assert len(res) == len(procs), "a rank ended without reporting"
    y = np.zeros_like(Ug)
    g = np.zeros_like(ref["gradUp"])
    mcs = 0.0
    xa = np.zeros_like(Ug)
    for rank, msg, idx, out, gg, speed, adv in res:
        assert msg == "ok", f"rank {rank}: {msg}"
        y[:, idx] = out
        g[:, :, idx] = gg
        mcs = max(mcs, speed)
        if adv is not None:
            xa[:, idx] = adv[0]
            assert adv[1] == pytest.approx(ref_adv[1], rel=1e-13) and adv[2] == pytest.approx(ref_adv[2], rel=1e-12)
            assert adv[3] == 0
    err = rel_maxnorm(y, ref["y"])
    print(world, "ranks: rel err", err)  # printed before any assertion: a failure then shows both legs
    if kind == "dry_air_nr":
        print("advance: rel err", rel_maxnorm(xa, ref_adv[0]))
        assert rel_maxnorm(xa, ref_adv[0]).max() < 1e-13
    assert err.max() < (5 * RHS_RTOL if kind in ("argon_2T", "axisym_2T", "axisym_slab") else RHS_RTOL)  # plasma: 1 % perturbations
    assert np.abs(g - ref["gradUp"]).max() < RHS_RTOL * np.abs(ref["gradUp"]).max()
    assert abs(mcs - ref["max_char_speed"]) < 1e-12 * mcs
