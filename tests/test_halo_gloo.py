"""World-size-2 rehearsal of the partitioned path on CPU (gloo): partitioning, shared-face tables,
the canonical-frame packing and the neighbour exchange of tps_amd.halo, checked with face-point
coordinates standing in for traces.  No GPU, no kernels."""
import ctypes as C
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from face_util import face_point_coords, gl_nodes, permute
from tps_amd import capi, cases, meshgen
from tps_amd.halo import HaloExchange


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _local_mesh(kind, rank, world):
    if kind == "slab":
        return meshgen.ogrid_cylinder_slab(3, 8, 3, rank, world), cases.cylinder_bcs(), np.array([0, 0, 2.0 * world])
    full = meshgen.scramble_orientations(meshgen.box_hex(4, 3, 3, warp=0.1), 9)
    owner = np.arange(full.num_elements) % world  # deliberately scattered partition: many neighbours faces
    return meshgen.partition(full, world, owner)[rank], [], np.array([1.0, 1.0, 1.0])


def _worker(rank, world, port, kind, q):
    try:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
        mesh, bcs, period = _local_mesh(kind, rank, world)
        dim, nlf, n = mesh.dim, 2 * mesh.dim, 4
        fn, fo, sslot, sorient = capi.face_tables(mesh, bcs)
        ns = len(sslot)
        assert ns == len(mesh.shared_neighbor_rank) and ns > 0
        pts = gl_nodes(n)
        per = n * n
        nfld = dim
        # pack like k_pack: canonical frame
        send = np.zeros((ns, nfld, per))
        mine = []
        for s in range(ns):
            e, f = divmod(int(sslot[s]), nlf)
            assert fn[e, f] == mesh.num_elements * nlf + s  # halo slot
            assert fo[e, f] == sorient[s]
            xy = face_point_coords(mesh, e, f, pts)  # [k][dim]
            mine.append(xy)
            for k in range(per):
                pk = permute(dim, int(sorient[s]), n, k % n, k // n)
                send[s, :, pk] = xy[k]
        recv = np.zeros_like(send)
        ranks = []
        offs = [0]
        for s in range(ns):
            r = int(mesh.shared_neighbor_rank[s])
            if not ranks or ranks[-1] != r:
                ranks.append(r)
                if s:
                    offs.append(s * nfld * per)
        offs.append(ns * nfld * per)
        halo = HaloExchange(host_buffers=True)
        ranks_c = (C.c_int * len(ranks))(*ranks)
        offs_c = (C.c_int64 * len(offs))(*offs)
        st = halo.callback(None, 0, send.ctypes.data, recv.ctypes.data, len(ranks), ranks_c, offs_c, offs_c, None)
        assert st == 0
        # what the consumer kernel reads: recv[halo slot][fld][permute(my->canonical, k)] at my point k
        for s in range(ns):
            for k in range(per):
                pk = permute(dim, int(sorient[s]), n, k % n, k // n)
                d = recv[s, :, pk] - mine[s][k]
                d -= np.round(d / np.where(period > 0, period, 1.0)) * period
                assert np.abs(d).max() < 1e-12, (rank, s, k, d)
        assert halo.bytes_sent == 8 * send.size
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as exc:  # pragma: no cover
        import traceback

        q.put((rank, traceback.format_exc()))


@pytest.mark.parametrize("kind", ["slab", "scattered"])
def test_two_rank_trace_exchange(kind):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, kind, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for r, msg in res:
        assert msg == "ok", f"rank {r}: {msg}"


def test_partition_covers_mesh():
    full = meshgen.ogrid_cylinder(3, 8, 4)
    parts = meshgen.partition(full, 3)
    assert sum(p.num_elements for p in parts) == full.num_elements
    assert sum(len(p.bdr_attributes) for p in parts) == len(full.bdr_attributes)
    ns = [len(p.shared_neighbor_rank) for p in parts]
    assert sum(ns) % 2 == 0 and min(ns) > 0


# ---- BASELINE.json configs[3] (cfg4): the 56 x 224 x 32 cylinder on EIGHT ranks, the partition bench.py --workload cfg4 builds ----
def _cfg4_worker(rank, world, port, q):
    try:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
        NR, NT, NZL, P, NEQ = 56, 224, 32 // world, 3, 5
        mesh = meshgen.ogrid_cylinder_slab(NR, NT, NZL, rank, world, span_local=2.0 * NZL / 16)
        bcs = cases.cylinder_bcs()
        assert mesh.num_elements == NR * NT * NZL == 401408 // world
        fn, fo, sslot, sorient = capi.face_tables(mesh, bcs)
        ns = len(sslot)
        # two spanwise neighbours, one plane of NR x NT faces each
        nbr = np.asarray(mesh.shared_neighbor_rank)
        assert ns == 2 * NR * NT and sorted(set(nbr.tolist())) == sorted({(rank - 1) % world, (rank + 1) % world})
        assert (np.diff(nbr) != 0).sum() == 1  # grouped by neighbour: one contiguous segment per neighbour
        e, f = np.divmod(np.asarray(sslot), 6)
        assert (fn[e, f] == mesh.num_elements * 6 + np.arange(ns)).all() and (fo[e, f] == np.asarray(sorient)).all()
        assert set(f.tolist()) == {4, 5}  # the z-faces
        # bytes per Mult and rank of the two exchange phases (tpsrhs.h: TA = traces of U and Up at the (p+1)^2 face nodes,
        # TB = the viscous normal flux of equations 1..neq-1 at the Q_f = 25 face points)
        n1, qf = P + 1, 25
        ta, tb = 2 * NEQ * n1 * n1 * 8, (NEQ - 1) * qf * 8
        per_mult = ns * (ta + tb)
        assert (ta, tb) == (1280, 800) and per_mult == 52_183_040  # 52 MB (SURVEY 8e estimated ~40 MB from ~9 000 shared faces)
        # a real exchange over gloo of the face-point coordinates of a sample of the shared faces, canonical frame
        n, dim = 4, 3
        pts = gl_nodes(n)
        per = n * n
        send = np.zeros((ns, dim, per))
        half = ns // 2
        J = list(range(0, half, 997)) + [half - 1]  # the same positions in both segments: what I send as face j of the
        sample = J + [j + half for j in J]         # segment towards a neighbour arrives as its face j of the segment from me
        mine = {}
        for s in sample:
            xy = face_point_coords(mesh, int(e[s]), int(f[s]), pts)
            mine[s] = xy
            for k in range(per):
                send[s, :, permute(dim, int(sorient[s]), n, k % n, k // n)] = xy[k]
        recv = np.zeros_like(send)
        ranks_c = (C.c_int * 2)(int(nbr[0]), int(nbr[-1]))
        offs_c = (C.c_int64 * 3)(0, half * dim * per, ns * dim * per)
        halo = HaloExchange(host_buffers=True)
        assert halo.callback(None, 0, send.ctypes.data, recv.ctypes.data, 2, ranks_c, offs_c, offs_c, None) == 0
        period = np.array([0.0, 0.0, 2.0 * 32 / 16])
        for s in sample:  # the neighbour samples the same faces (the shared planes are listed in the same order on both sides)
            for k in range(per):
                d = recv[s, :, permute(dim, int(sorient[s]), n, k % n, k // n)] - mine[s][k]
                d[2] -= np.round(d[2] / period[2]) * period[2]
                assert np.abs(d).max() < 1e-12, (rank, s, k, d)
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback

        q.put((rank, traceback.format_exc()))


def test_cfg4_eight_rank_slab_partition():
    """configs[3] at its own size and rank count: 8 ranks x (56 x 224 x 4) hexes -- face tables, halo slots, segment
    offsets and bytes per Mult of every rank, and an exchange of shared-face coordinates over gloo (no GPU, no kernels)."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cfg4_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for r, msg in res:
        assert msg == "ok", f"rank {r}: {msg}"
