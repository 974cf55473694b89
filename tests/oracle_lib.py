"""Loader of the CPU oracle (test infrastructure).  Imported by tests/, smoke() and bench.py's
cpu_baseline leg only."""
import ctypes as C
import os
import subprocess

import numpy as np

from tps_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "_build", "libtpsoracle.so")
_lib = None
_dp = C.POINTER(C.c_double)


def build():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            build()
        L = C.CDLL(ORACLE_SO)
        vp = C.c_void_p
        L.tpsoracle_create.argtypes = [C.POINTER(capi.Mesh), C.POINTER(capi.Disc), C.POINTER(capi.Physics), C.c_int,
                                       C.POINTER(capi.BC), C.POINTER(vp)]
        L.tpsoracle_last_error.restype = C.c_char_p
        L.tpsoracle_num_dofs.restype = C.c_int64
        L.tpsoracle_num_dofs.argtypes = [vp]
        L.tpsoracle_num_equation.argtypes = [vp]
        L.tpsoracle_destroy.argtypes = [vp]
        L.tpsoracle_mult.argtypes = [vp, _dp, _dp, C.c_double, _dp]
        L.tpsoracle_compute_gradients.argtypes = [vp, _dp]
        L.tpsoracle_update_primitives.argtypes = [vp, _dp]
        L.tpsoracle_get_primitives.argtypes = [vp, _dp]
        L.tpsoracle_get_gradients.argtypes = [vp, _dp]
        L.tpsoracle_node_coords.argtypes = [vp, _dp]
        L.tpsoracle_advance.argtypes = [vp, _dp, _dp, _dp, C.c_int, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_int64)]
        L.tpsoracle_set_dt.argtypes = [vp, C.c_double]
        L.tpsoracle_get_boundary_state.argtypes = [vp, C.c_int, _dp, _dp]
        L.tpsoracle_set_forcing.argtypes = [vp, C.POINTER(capi.Forcing)]
        L.tpsoracle_set_joule_heating.argtypes = [vp, _dp]
        L.tpsoracle_l2_norm.restype = C.c_double
        L.tpsoracle_l2_norm.argtypes = [vp, _dp, _dp]
        L.tpsoracle_integral.restype = C.c_double
        L.tpsoracle_integral.argtypes = [vp, _dp]
        L.tpsoracle_point_pressure.restype = C.c_double
        L.tpsoracle_point_max_char_speed.restype = C.c_double
        for name in ("tpsoracle_point_prim", "tpsoracle_point_cons", "tpsoracle_point_convective_flux"):
            getattr(L, name).argtypes = [vp, _dp, _dp]
        L.tpsoracle_point_pressure.argtypes = [vp, _dp]
        L.tpsoracle_point_max_char_speed.argtypes = [vp, _dp]
        L.tpsoracle_point_viscous_flux.argtypes = [vp, _dp, _dp, C.c_double, _dp]
        L.tpsoracle_point_viscous_flux_at.argtypes = [vp, _dp, _dp, _dp, C.c_double, _dp]
        L.tpsoracle_element_sizes.argtypes = [vp, _dp]
        L.tpsoracle_point_bdr_viscous_flux.argtypes = [vp, _dp, _dp, C.c_double, _dp, _dp, C.POINTER(C.c_int), _dp]
        L.tpsoracle_point_lf.argtypes = [vp, _dp, _dp, _dp, _dp]
        L.tpsoracle_point_roe.argtypes = [vp, _dp, _dp, _dp, _dp]
        L.tpsoracle_point_bdr_flux.argtypes = [vp, C.c_int, _dp, _dp, _dp, C.c_double, _dp]
        L.tpsoracle_point_flux_transport.argtypes = [vp, _dp, _dp, _dp, _dp]
        L.tpsoracle_point_source.argtypes = [vp, _dp, _dp, _dp, _dp]
        L.tpsoracle_rk4_step.argtypes = [vp, _dp, _dp, C.c_double, _dp, C.POINTER(C.c_int64)]
        L.tpsoracle_point_source_transport.argtypes = [vp, _dp, _dp, _dp, _dp, _dp, _dp, _dp]
        L.tpsoracle_collision_integral.restype = C.c_double
        L.tpsoracle_collision_integral.argtypes = [C.c_int, C.c_double]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(_dp)


class Oracle:
    """CPU restatement of RHSoperator (oracle/tps_oracle.cpp)."""

    def __init__(self, host_mesh, disc, physics, bcs=(), threads=None):
        L = lib()
        if threads is not None:
            L.tpsoracle_set_threads(int(threads))
        self._margs = capi.MeshArgs(host_mesh)
        self.disc, self.physics = disc, physics
        self._bcs = (capi.BC * max(1, len(bcs)))(*bcs)
        h = C.c_void_p()
        st = L.tpsoracle_create(C.byref(self._margs.c), C.byref(disc), C.byref(physics), len(bcs), self._bcs,
                                C.byref(h))
        if st != 0:
            raise RuntimeError("oracle: " + L.tpsoracle_last_error().decode())
        self.h = h
        self.dim = host_mesh.dim
        self.ndofs = int(L.tpsoracle_num_dofs(h))
        self.neq = int(L.tpsoracle_num_equation(h))

    def __del__(self):
        if getattr(self, "h", None):
            lib().tpsoracle_destroy(self.h)
            self.h = None

    def node_coords(self):
        out = np.zeros((self.dim, self.ndofs))
        lib().tpsoracle_node_coords(self.h, _p(out))
        return out

    def boundary_state(self, attr):
        """(boundaryU [points, neq], meanUp [neq]) of a non-reflecting patch"""
        n = lib().tpsoracle_get_boundary_state(self.h, int(attr), None, None)
        bu = np.zeros((n, self.neq))
        mean = np.zeros(self.neq)
        lib().tpsoracle_get_boundary_state(self.h, int(attr), _p(bu), _p(mean))
        return bu, mean

    def set_dt(self, dt):
        lib().tpsoracle_set_dt(self.h, float(dt))

    def set_forcing(self, forcing):
        st = lib().tpsoracle_set_forcing(self.h, C.byref(forcing) if forcing is not None else None)
        if st != 0:
            raise RuntimeError("oracle: " + lib().tpsoracle_last_error().decode())

    def set_mixing_length(self, distance, max_mixing_length=0.0, pr_ratio=1.0, lewis=1.0, bulk_multiplier=0.0):
        """MixingLengthTransport around the molecular transport; distance: host array of NDofs entries or None"""
        L = lib()
        L.tpsoracle_set_mixing_length.argtypes = [C.c_void_p, _dp, C.POINTER(capi.MixingLength)]
        if distance is None:
            L.tpsoracle_set_mixing_length(self.h, None, None)
            return
        d = np.ascontiguousarray(distance, dtype=np.float64)
        assert d.size == self.ndofs
        prm = capi.MixingLength(float(max_mixing_length), float(pr_ratio), float(lewis), float(bulk_multiplier))
        L.tpsoracle_set_mixing_length(self.h, _p(d), C.byref(prm))

    def set_joule_heating(self, jh):
        if jh is None:
            lib().tpsoracle_set_joule_heating(self.h, None)
        else:
            jh = np.ascontiguousarray(jh, dtype=np.float64)
            assert jh.size == self.ndofs
            lib().tpsoracle_set_joule_heating(self.h, _p(jh))

    def mult(self, x, time=0.0):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.zeros_like(x)
        mcs = C.c_double(0.0)
        st = lib().tpsoracle_mult(self.h, _p(x), _p(y), float(time), C.byref(mcs))
        if st != 0:
            raise RuntimeError("oracle: " + lib().tpsoracle_last_error().decode())
        self.max_char_speed = mcs.value
        return y

    def compute_gradients(self, up=None):
        if up is not None:
            up = np.ascontiguousarray(up, dtype=np.float64)
        st = lib().tpsoracle_compute_gradients(self.h, _p(up) if up is not None else None)
        if st != 0:
            raise RuntimeError("oracle: " + lib().tpsoracle_last_error().decode())
        return self.gradients()

    def primitives(self):
        out = np.zeros((self.neq, self.ndofs))
        lib().tpsoracle_get_primitives(self.h, _p(out))
        return out

    def plasma_conductivity(self, x):
        """plasma_conductivity_ of SourceTerm for the state x"""
        out = np.zeros(int(self.ndofs))
        L = lib()
        L.tpsoracle_get_plasma_conductivity.argtypes = [C.c_void_p, _dp, _dp]
        xx = np.ascontiguousarray(x, dtype=np.float64)
        if L.tpsoracle_get_plasma_conductivity(self.h, _p(xx), _p(out)) != 0:
            raise RuntimeError("oracle: " + L.tpsoracle_last_error().decode())
        return out

    def gradients(self):
        out = np.zeros((self.dim, self.neq, self.ndofs))
        lib().tpsoracle_get_gradients(self.h, _p(out))
        return out

    def l2_norm(self, a, b=None):
        a = np.ascontiguousarray(a, dtype=np.float64)
        if b is not None:
            b = np.ascontiguousarray(b, dtype=np.float64)
        return lib().tpsoracle_l2_norm(self.h, _p(a), _p(b) if b is not None else None)

    def integral(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        return lib().tpsoracle_integral(self.h, _p(a))

    # ---- point-wise ----
    def prim(self, state):
        s = np.ascontiguousarray(state, dtype=np.float64)
        out = np.zeros(self.neq)
        lib().tpsoracle_point_prim(self.h, _p(s), _p(out))
        return out

    def cons(self, prim):
        s = np.ascontiguousarray(prim, dtype=np.float64)
        out = np.zeros(self.neq)
        lib().tpsoracle_point_cons(self.h, _p(s), _p(out))
        return out

    def pressure(self, state):
        s = np.ascontiguousarray(state, dtype=np.float64)
        return lib().tpsoracle_point_pressure(self.h, _p(s))

    def max_char_speed_point(self, state):
        s = np.ascontiguousarray(state, dtype=np.float64)
        return lib().tpsoracle_point_max_char_speed(self.h, _p(s))

    def convective_flux(self, state):
        s = np.ascontiguousarray(state, dtype=np.float64)
        out = np.zeros(self.neq * self.dim)
        lib().tpsoracle_point_convective_flux(self.h, _p(s), _p(out))
        return out.reshape(self.dim, self.neq)

    def viscous_flux(self, state, grad, radius=-1.0):
        s = np.ascontiguousarray(state, dtype=np.float64)
        g = np.ascontiguousarray(grad, dtype=np.float64)  # (dim, neq): gradUp[eq + d*neq]
        out = np.zeros(self.neq * self.dim)
        lib().tpsoracle_point_viscous_flux(self.h, _p(s), _p(g), float(radius), _p(out))
        return out.reshape(self.dim, self.neq)

    def viscous_flux_at(self, state, grad, x, delta):
        """F_v[eq + d*neq] at position x in an element of grid scale delta (sub-grid scale models, viscous sponge)"""
        s = np.ascontiguousarray(state, dtype=np.float64)
        g = np.ascontiguousarray(grad, dtype=np.float64)
        xx = np.ascontiguousarray(x, dtype=np.float64)
        out = np.zeros(self.neq * self.dim)
        lib().tpsoracle_point_viscous_flux_at(self.h, _p(s), _p(g), _p(xx), float(delta), _p(out))
        return out

    def element_sizes(self):
        out = np.zeros(int(self.ndofs))
        n = lib().tpsoracle_element_sizes(self.h, _p(out))
        return out[:n]

    def bdr_viscous_flux(self, state, grad, normal, prim_flux=None, prim_flux_idxs=None, radius=-1.0):
        s = np.ascontiguousarray(state, dtype=np.float64)
        g = np.ascontiguousarray(grad, dtype=np.float64)
        n = np.zeros(3)
        n[: len(normal)] = normal
        pf = np.zeros(capi.MAXEQUATIONS) if prim_flux is None else np.ascontiguousarray(prim_flux, dtype=np.float64)
        pi = np.zeros(capi.MAXEQUATIONS, dtype=np.int32) if prim_flux_idxs is None else np.ascontiguousarray(
            prim_flux_idxs, dtype=np.int32)
        out = np.zeros(self.neq)
        lib().tpsoracle_point_bdr_viscous_flux(self.h, _p(s), _p(g), float(radius), _p(n), _p(pf),
                                               pi.ctypes.data_as(C.POINTER(C.c_int)), _p(out))
        return out

    def lf(self, s1, s2, nor):
        s1 = np.ascontiguousarray(s1, dtype=np.float64)
        s2 = np.ascontiguousarray(s2, dtype=np.float64)
        n = np.zeros(3)
        n[: len(nor)] = nor
        out = np.zeros(self.neq)
        lib().tpsoracle_point_lf(self.h, _p(s1), _p(s2), _p(n), _p(out))
        return out

    def roe(self, s1, s2, nor):
        s1 = np.ascontiguousarray(s1, dtype=np.float64)
        s2 = np.ascontiguousarray(s2, dtype=np.float64)
        n = np.zeros(3)
        n[: len(nor)] = nor
        out = np.zeros(self.neq)
        lib().tpsoracle_point_roe(self.h, _p(s1), _p(s2), _p(n), _p(out))
        return out

    def bdr_flux(self, attr, nor, state, grad, radius=-1.0):
        s = np.ascontiguousarray(state, dtype=np.float64)
        g = np.ascontiguousarray(grad, dtype=np.float64)
        n = np.zeros(3)
        n[: len(nor)] = nor
        out = np.zeros(self.neq)
        st = lib().tpsoracle_point_bdr_flux(self.h, int(attr), _p(n), _p(s), _p(g), float(radius), _p(out))
        if st != 0:
            raise RuntimeError("oracle: " + lib().tpsoracle_last_error().decode())
        return out

    def rk4_step(self, x, time, dt):
        """-> (new x, new time, max_char_speed of the last stage, NaN count)"""
        xx = np.ascontiguousarray(x, dtype=np.float64).copy()
        t = C.c_double(time)
        speed = C.c_double(0.0)
        bad = C.c_int64(0)
        st = lib().tpsoracle_rk4_step(self.h, _p(xx), C.byref(t), float(dt), C.byref(speed), C.byref(bad))
        if st != 0:
            raise RuntimeError("oracle: " + lib().tpsoracle_last_error().decode())
        return xx, t.value, speed.value, bad.value

    def advance(self, x, time, dt, num_steps, constant_dt=True, cfl=0.0, hmin=0.0):
        """-> (new x, time, next dt, NaN count)"""
        xx = np.ascontiguousarray(x, dtype=np.float64).copy()
        t, d, bad = C.c_double(time), C.c_double(dt), C.c_int64(0)
        st = lib().tpsoracle_advance(self.h, _p(xx), C.byref(t), C.byref(d), int(num_steps), 1 if constant_dt else 0,
                                     float(cfl), float(hmin), C.byref(bad))
        if st != 0:
            raise RuntimeError("oracle: " + lib().tpsoracle_last_error().decode())
        return xx, t.value, d.value, bad.value

    def flux_transport(self, state, grad):
        """(viscosity, bulk, k_heavy, k_electron), diffusion velocities [sp + d*nsp]"""
        buf = np.zeros(4)
        V = np.zeros(3 * 8)
        lib().tpsoracle_point_flux_transport(self.h, _p(np.ascontiguousarray(state, dtype=np.float64)),
                                             _p(np.ascontiguousarray(grad, dtype=np.float64)), _p(buf), _p(V))
        return buf, V

    def source_transport(self, state, prim, grad):
        """electric conductivity, momentum-transfer frequencies, diffusion velocities, number densities"""
        g, sp, V, n = np.zeros(8), np.zeros(8), np.zeros(24), np.zeros(8)
        lib().tpsoracle_point_source_transport(self.h, _p(np.ascontiguousarray(state, dtype=np.float64)),
                                               _p(np.ascontiguousarray(prim, dtype=np.float64)),
                                               _p(np.ascontiguousarray(grad, dtype=np.float64)), _p(g), _p(sp), _p(V), _p(n))
        return g[0], sp, V, n

    def source(self, state, prim, grad):
        out = np.zeros(self.neq)
        st = lib().tpsoracle_point_source(self.h, _p(np.ascontiguousarray(state, dtype=np.float64)),
                                          _p(np.ascontiguousarray(prim, dtype=np.float64)),
                                          _p(np.ascontiguousarray(grad, dtype=np.float64)), _p(out))
        if st != 0:
            raise RuntimeError("oracle: no source term for this physics")
        return out


def collision_integral(name, x):
    ids = {"att11": 0, "att12": 1, "att13": 2, "att14": 3, "att15": 4, "rep22": 5, "rep23": 6, "rep24": 7,
           "ArAr22": 8, "ArAr1P11": 9, "eAr11": 10, "eAr12": 11, "eAr13": 12, "eAr14": 13, "eAr15": 14}
    return lib().tpsoracle_collision_integral(ids[name], float(x))



def table_eval(table, x):
    """LinearTable of the oracle on host arrays -> (values, interval indices)"""
    x = np.ascontiguousarray(x, dtype=np.float64)
    f = np.empty_like(x)
    idx = np.empty(x.shape, dtype=np.int32)
    L = lib()
    L.tpsoracle_table_eval.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    st = L.tpsoracle_table_eval(C.byref(table), x.size, x.ctypes.data, f.ctypes.data, idx.ctypes.data)
    assert st == 0, L.tpsoracle_last_error().decode()
    return f, idx
