"""Python image of the reference's ``RHSoperator`` (``src/rhs_operator.hpp:59-235``) over the C ABI.

The class keeps the reference's method names and argument meaning -- ``Mult(x, y)``,
``updatePrimitives``/``updateGradients``, ``getGradients`` -- so that tests read like
``utils/compute_rhs.cpp:60-102`` and ``test/test_gradient.cpp:159-162``.  ``x`` and ``y`` are
torch CUDA tensors (float64, ``num_equation * NDofs``, byNODES); torch is used only to own device
memory and streams.  Everything is computed by ``libtpsrhs.so``; a missing library or a missing
GPU raises -- there is no CPU path.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import capi


class TpsRhsError(RuntimeError):
    def __init__(self, status, where):
        lib = capi.load()
        self.status = status
        msg = lib.tpsrhs_last_error().decode()
        super().__init__(f"{where}: {lib.tpsrhs_status_string(status).decode()}: {msg}")


class RHSoperator:
    """``RHSoperator : mfem::TimeDependentOperator`` -- one instance per rank / per GPU.

    Parameters mirror what ``M2ulPhyS::initVariables`` hands the reference constructor
    (``src/M2ulPhyS.cpp:745-749``): the (local) mesh, the discretisation, the physics parameter
    blocks, the boundary conditions.  ``halo`` is the neighbour-exchange hook used on partitioned
    meshes (see :mod:`tps_amd.halo`).
    """

    def __init__(self, host_mesh, disc, physics, bcs=(), device=0, halo=None, stream=None):
        self._lib = capi.load()
        if not torch.cuda.is_available():
            raise RuntimeError("tps_amd.RHSoperator needs a HIP device (torch.cuda.is_available() is False)")
        self.device = torch.device("cuda", device)
        self._margs = capi.MeshArgs(host_mesh)
        self._disc, self._physics = disc, physics
        self._bcs = (capi.BC * max(1, len(bcs)))(*bcs)
        rt = capi.Runtime()
        rt.device = device
        st = stream if stream is not None else torch.cuda.current_stream(self.device)
        self._stream = st
        rt.stream = C.c_void_p(st.cuda_stream)
        self._halo = halo
        if halo is not None and hasattr(halo, "c_halo"):  # native exchange (tps_amd.halo_rccl): C function pointers
            rt.halo, rt.halo_ctx = halo.c_halo, halo.ctx
            rt.reduce, rt.reduce_ctx = halo.c_reduce, halo.ctx
        elif halo is not None:  # Python hook (tps_amd.halo): gloo rehearsals and tests
            self._halo_cb = capi.HALO_FN(halo.callback)
            rt.halo = self._halo_cb
            if hasattr(halo, "reduce_callback"):
                self._reduce_cb = capi.REDUCE_FN(halo.reduce_callback)
                rt.reduce = self._reduce_cb
        self._rt = rt
        h = C.c_void_p()
        st_code = self._lib.tpsrhs_create(C.byref(self._margs.c), C.byref(disc), C.byref(physics), len(bcs),
                                          self._bcs, C.byref(rt), C.byref(h))
        if st_code != 0:
            raise TpsRhsError(st_code, "tpsrhs_create")
        self._h = h
        self.dim = host_mesh.dim
        self.num_equation = int(self._lib.tpsrhs_num_equation(h))
        self.NDofs = int(self._lib.tpsrhs_num_dofs(h))
        self.max_char_speed = 0.0
        self._time = 0.0

    # -- mfem::Operator / TimeDependentOperator surface ------------------------------------
    def Height(self) -> int:
        return int(self._lib.tpsrhs_height(self._h))

    def SetTime(self, t: float):
        self._time = float(t)

    def GetTime(self) -> float:
        return self._time

    def Mult(self, x: torch.Tensor, y: torch.Tensor, want_max_char_speed: bool = False):
        """``y = RHS(x)`` (``src/rhs_operator.cpp:343-464``).  Asynchronous on the operator's stream
        unless ``want_max_char_speed`` (the reference's ``max_char_speed`` side effect)."""
        self._check(x)
        self._check(y)
        mcs = C.c_double(0.0)
        st = self._lib.tpsrhs_mult(self._h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), self._time,
                                   C.byref(mcs) if want_max_char_speed else None)
        if st != 0:
            raise TpsRhsError(st, "tpsrhs_mult")
        if want_max_char_speed:
            self.max_char_speed = mcs.value

    def updateGradients(self, x: torch.Tensor):
        self._check(x)
        st = self._lib.tpsrhs_update_gradients(self._h, C.c_void_p(x.data_ptr()))
        if st != 0:
            raise TpsRhsError(st, "tpsrhs_update_gradients")

    def getPrimitives(self) -> torch.Tensor:
        out = torch.empty(self.num_equation * self.NDofs, dtype=torch.float64, device=self.device)
        st = self._lib.tpsrhs_get_primitives(self._h, C.c_void_p(out.data_ptr()))
        if st != 0:
            raise TpsRhsError(st, "tpsrhs_get_primitives")
        return out.view(self.num_equation, self.NDofs)

    def getGradients(self) -> torch.Tensor:
        out = torch.empty(self.dim * self.num_equation * self.NDofs, dtype=torch.float64, device=self.device)
        st = self._lib.tpsrhs_get_gradients(self._h, C.c_void_p(out.data_ptr()))
        if st != 0:
            raise TpsRhsError(st, "tpsrhs_get_gradients")
        return out.view(self.dim, self.num_equation, self.NDofs)

    def getPlasmaConductivity(self, x: torch.Tensor) -> torch.Tensor:
        """``plasma_conductivity_`` of ``SourceTerm`` (``src/source_term.cpp:125-199``): sigma at the nodes of the state
        ``x`` -- what the EM solver of the coupled torch runs reads (mixtures and the table gas)."""
        self._check(x)
        out = torch.empty(self.NDofs, dtype=torch.float64, device=self.device)
        st = self._lib.tpsrhs_get_plasma_conductivity(self._h, C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()))
        if st != 0:
            raise TpsRhsError(st, "tpsrhs_get_plasma_conductivity")
        return out

    # -- measurement helpers -----------------------------------------------------------------
    def enable_kernel_timing(self, on=True):
        self._lib.tpsrhs_enable_kernel_timing(self._h, 1 if on else 0)

    def kernel_times(self):
        names = (C.c_char_p * 8)()
        ms = (C.c_double * 8)()
        n = self._lib.tpsrhs_kernel_times(self._h, 8, names, ms)
        return {names[i].decode(): ms[i] for i in range(n)}

    def mult_times(self):
        """Device milliseconds of each ``Mult`` since timing was enabled (at most the last 128)."""
        ms = (C.c_double * 128)()
        n = self._lib.tpsrhs_mult_times(self._h, 128, ms)
        return [ms[i] for i in range(n)]

    def rk4_step(self, x: torch.Tensor, time: float, dt: float, want_max_char_speed=False, want_nan_count=False):
        """One explicit RK4 step, ``x`` updated in place; returns the new time (the role of
        ``timeIntegrator->Step(*U, time, dt); Check_NAN(); Check_Undershoot();`` in
        ``M2ulPhyS::solveStep``, ``src/M2ulPhyS.cpp:2004-2008``)."""
        self._check(x)
        t = C.c_double(time)
        speed = C.c_double(0.0)
        bad = C.c_int64(0)
        st = self._lib.tpsrhs_rk4_step(self._h, C.c_void_p(x.data_ptr()), C.byref(t), float(dt),
                                       C.byref(speed) if want_max_char_speed else None,
                                       C.byref(bad) if want_nan_count else None)
        if st != 0:
            raise RuntimeError(f"tpsrhs_rk4_step: {capi.STATUS.get(st, st)}: {self._lib.tpsrhs_last_error().decode()}")
        if want_max_char_speed:
            self.max_char_speed = speed.value
        if want_nan_count:
            self.nan_count = bad.value
        return t.value

    def advance(self, x: torch.Tensor, time: float, dt: float, num_steps: int, constant_dt=True, cfl=0.0, hmin=0.0):
        """``num_steps`` times ``M2ulPhyS::solveStep`` (``src/M2ulPhyS.cpp:2004-2019``) with dt, time and the NaN
        census on the device; returns ``(time, next dt, NaN count)`` after ONE synchronisation at the end."""
        self._check(x)
        t, d, bad = C.c_double(time), C.c_double(dt), C.c_int64(0)
        st = self._lib.tpsrhs_advance(self._h, C.c_void_p(x.data_ptr()), C.byref(t), C.byref(d), int(num_steps),
                                      1 if constant_dt else 0, float(cfl), float(hmin), C.byref(bad))
        if st != 0:
            raise TpsRhsError(st, "tpsrhs_advance")
        return t.value, d.value, bad.value

    def setDt(self, dt: float):
        """The ``dt`` the non-reflecting boundary conditions advance their boundary state with in every
        ``Mult`` (the reference's ``BoundaryCondition::dt`` is a reference to ``M2ulPhyS::dt``)."""
        st = self._lib.tpsrhs_set_dt(self._h, float(dt))
        if st != 0:
            raise TpsRhsError(st, "tpsrhs_set_dt")

    def setForcing(self, forcing):
        """ConstantPressureGradient / SpongeZone / HeatSource of the reference's ``forcing`` array
        (``src/rhs_operator.cpp:101-123``); ``forcing``: :class:`tps_amd.capi.Forcing` or ``None``."""
        st = self._lib.tpsrhs_set_forcing(self._h, C.byref(forcing) if forcing is not None else None)
        if st != 0:
            raise TpsRhsError(st, "tpsrhs_set_forcing")

    def setJouleHeating(self, joule_heating):
        """The ``joule_heating_`` grid function of ``JouleHeating`` (``src/forcing_terms.cpp:443-471``): a
        float64 CUDA tensor of NDofs entries the operator reads at every ``Mult`` (kept alive here), or ``None``."""
        if joule_heating is not None:
            if (joule_heating.dtype != torch.float64 or not joule_heating.is_cuda or not joule_heating.is_contiguous()
                    or joule_heating.numel() != self.NDofs):
                raise ValueError("expected a contiguous float64 CUDA tensor of NDofs entries")
        self._joule = joule_heating
        st = self._lib.tpsrhs_set_joule_heating(
            self._h, C.c_void_p(joule_heating.data_ptr()) if joule_heating is not None else None)
        if st != 0:
            raise TpsRhsError(st, "tpsrhs_set_joule_heating")

    def setMixingLength(self, distance, max_mixing_length=0.0, pr_ratio=1.0, lewis=1.0, bulk_multiplier=0.0):
        """``MixingLengthTransport`` (``src/mixing_length_transport.cpp``, ``[flow] useMixingLength``): ``distance`` =
        the wall-distance grid function, a float64 CUDA tensor of NDofs entries read at every ``Mult`` (kept alive
        here), or ``None`` to switch the model off."""
        if distance is not None:
            if (distance.dtype != torch.float64 or not distance.is_cuda or not distance.is_contiguous()
                    or distance.numel() != self.NDofs):
                raise ValueError("expected a contiguous float64 CUDA tensor of NDofs entries")
        self._distance = distance
        prm = capi.MixingLength(float(max_mixing_length), float(pr_ratio), float(lewis), float(bulk_multiplier))
        st = self._lib.tpsrhs_set_mixing_length(self._h, C.c_void_p(distance.data_ptr()) if distance is not None else None,
                                                C.byref(prm))
        if st != 0:
            raise TpsRhsError(st, "tpsrhs_set_mixing_length")

    def kernel_bytes(self):
        names = (C.c_char_p * 8)()
        b = (C.c_double * 8)()
        n = self._lib.tpsrhs_kernel_bytes(self._h, 8, names, b)
        return {names[i].decode(): b[i] for i in range(n)}

    def _check(self, t: torch.Tensor):
        if t.dtype != torch.float64 or not t.is_cuda or not t.is_contiguous() or t.numel() != self.Height():
            raise ValueError("expected a contiguous float64 CUDA tensor of num_equation*NDofs entries")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.tpsrhs_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def node_coordinates(host_mesh, order: int, basis_type: int = 0) -> np.ndarray:
    """Physical coordinates of the DG nodes, ``(dim, NDofs)``: the role of
    ``mesh->GetNodes(*coordsDof)`` (``src/rhs_operator.cpp:139-142``) for a Gauss-Legendre (``basis_type`` 0) or
    Gauss-Lobatto (1) nodal basis on order-1 geometry.  Host-side input generation only."""
    dim = host_mesh.dim
    n1 = order + 1
    if basis_type == 0:
        x, _ = np.polynomial.legendre.leggauss(n1)
    else:  # end points and the roots of P'_{n1-1}
        inner = np.polynomial.legendre.Legendre.basis(n1 - 1).deriv().roots() if n1 > 2 else np.array([])
        x = np.concatenate([[-1.0], np.sort(inner.real), [1.0]])
    x = 0.5 * (x + 1.0)
    ex = host_mesh.elem_coords  # (ne, 2^dim, dim), MFEM order
    if dim == 3:
        corners = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)]
        k, j, i = np.meshgrid(x, x, x, indexing="ij")
        xi = [i.ravel(), j.ravel(), k.ravel()]
    else:
        corners = [(0, 0), (1, 0), (1, 1), (0, 1)]
        j, i = np.meshgrid(x, x, indexing="ij")
        xi = [i.ravel(), j.ravel()]
    out = np.zeros((dim, host_mesh.num_elements, xi[0].size))
    for v, c in enumerate(corners):
        shp = np.ones_like(xi[0])
        for d in range(dim):
            shp = shp * (xi[d] if c[d] else 1.0 - xi[d])
        for d in range(dim):
            out[d] += ex[:, v, d][:, None] * shp[None, :]
    return out.reshape(dim, -1)
