"""ctypes front end of ``oracle/c/tz_oracle.c`` -- ORACLE / TEST INFRASTRUCTURE (see oracle/__init__.py).

Takes the UNSCALED parametric QP (any object with the attributes of ``tzddpc_amd.builder.ParametricQP``)
and runs the plain-C restatement of the per-step numeric path on the host cores.  Used by
``tests/`` as the full-size checker and by ``bench.py`` as ``cpu_baseline`` (kind "port").
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "_build", "libtz_oracle.so")
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


class Desc(C.Structure):
    _fields_ = [("n", C.c_int32), ("m", C.c_int32), ("N", C.c_int32), ("nz", C.c_int32), ("nc", C.c_int32), ("ntheta", C.c_int32),
                ("npar", C.c_int32), ("P", _dp), ("A", _dp), ("q0", _dp), ("Qt", _dp), ("l0", _dp), ("Lt", _dp), ("u0", _dp), ("Ut", _dp),
                ("f0", _dp), ("Ft", _dp), ("pl", _dp), ("pu", _dp), ("r0", C.c_double), ("r1", _dp), ("R2", _dp), ("Phi", _dp), ("Gam", _dp),
                ("CK", _dp), ("DK", _dp), ("K", _dp), ("pmax", C.c_int32), ("absCK", _dp), ("absKCK", _dp), ("power", _ip),
                ("max_iter", C.c_int32), ("tol", C.c_double), ("reg", C.c_double), ("step_frac", C.c_double),
                ("warm_floor", C.c_double), ("warm_gain", C.c_double), ("warm_cap", C.c_double), ("mu_tol", C.c_double), ("res_tol", C.c_double), ("aff_thr", C.c_double), ("aff_mu", C.c_double),
                ("shift_var", _ip), ("shift_row", _ip), ("shift_policy", C.c_int32), ("shift_quiet", C.c_int32), ("start_xbar0", _dp)]


def _cpu_tag() -> str:
    """Identity of the host CPU: the library is built with -march=native, so a copy built elsewhere must not be loaded blindly."""
    try:
        import hashlib
        with open("/proc/cpuinfo") as f:
            txt = f.read()
        model = next((ln.split(":", 1)[1].strip() for ln in txt.splitlines() if ln.startswith("model name")), "?")
        flags = next((ln.split(":", 1)[1].strip() for ln in txt.splitlines() if ln.startswith("flags")), "")
        return model + " " + hashlib.sha1(flags.encode()).hexdigest()[:12]
    except OSError:
        return "unknown"


def build(force: bool = False) -> str:
    """Rebuild when the library is missing, was built on another CPU (-march=native) or from other sources (content hash of the
    C file and the Makefile: file times do not survive checkouts and snapshot copies)."""
    import hashlib
    cdir = os.path.join(_HERE, "c")
    h = hashlib.sha256()
    for name in ("tz_oracle.c", "Makefile"):
        with open(os.path.join(cdir, name), "rb") as fh:
            h.update(fh.read())
    want = _cpu_tag() + "\n" + h.hexdigest()
    tag = os.path.join(_HERE, "_build", "cpu.txt")
    current = os.path.exists(LIB) and os.path.exists(tag) and open(tag).read() == want
    if force or not current:
        subprocess.run(["make", "-s", "-B", "-C", cdir], check=True)
        with open(tag, "w") as f:
            f.write(want)
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        alt = os.environ.get("TZ_ORACLE_LIB")            # a sanitizer build of the same source (tests/test_oracle_sanitizers.py)
        if not alt:
            build()                  # no-op when the library is current and was built on this CPU
        _lib = C.CDLL(alt or LIB)
        _lib.tzo_max_threads.restype = C.c_int
    return _lib


class COracle:
    def __init__(self, qp, max_iter=40, tol=1e-10, reg=1e-12, step_frac=0.99999, warm_floor=1e-8, warm_gain=1.0, warm_cap=1e300, mu_factor=1e-3, res_factor=100.0, aff_thr=0.99, aff_mu=1e-3, shift_policy=0, shift_maps=None, shift_quiet=16, stored_start=None):
        self.qp = qp
        self._keep = []
        d = Desc()
        f = lambda a: self._pin(np.ascontiguousarray(a, dtype=np.float64))
        d.n, d.m, d.N, d.nz, d.nc, d.ntheta, d.npar = qp.n, qp.m, qp.N, qp.nz, qp.nc, qp.ntheta, len(qp.f0)
        for name, arr in dict(P=qp.P, A=qp.A, q0=qp.q0, Qt=qp.Qt, l0=qp.l0, Lt=qp.Lt, u0=qp.u0, Ut=qp.Ut, r1=qp.r1, R2=qp.R2,
                              Phi=qp.Phi, Gam=qp.Gam, CK=qp.tube.CK, DK=qp.tube.DK, K=qp.tube.K, absCK=qp.tube.absCKpow,
                              absKCK=qp.tube.absKCKpow).items():
            setattr(d, name, f(arr).ctypes.data_as(_dp))
        for name, arr in dict(f0=qp.f0, Ft=qp.Ft, pl=qp.pl, pu=qp.pu).items():
            a = arr if np.size(arr) else np.zeros(1)
            setattr(d, name, f(a).ctypes.data_as(_dp))
        pw = self._pin(np.ascontiguousarray(qp.tube.power, dtype=np.int32))
        d.power = pw.ctypes.data_as(_ip)
        d.r0 = float(qp.r0); d.pmax = int(qp.tube.pmax)
        d.max_iter, d.tol, d.reg, d.step_frac = int(max_iter), float(tol), float(reg), float(step_frac)
        d.mu_tol = float(tol) * float(mu_factor)
        d.res_tol = float(tol) * float(res_factor)
        d.aff_thr, d.aff_mu = float(aff_thr), float(aff_mu)
        d.warm_floor, d.warm_gain = float(warm_floor), float(warm_gain)       # warm_floor = 0: every closed-loop step starts cold
        d.shift_quiet = int(shift_quiet)
        d.warm_cap = float(min(warm_cap, 1e300))
        if shift_policy:                    # same receding-horizon shift of the warm start as the device (TZDDPC.warm_shift_policy):
            if shift_maps is None:          # the maps (source variable / source two-sided row) are handed in by the caller
                raise ValueError("shift_policy != 0 needs shift_maps=(source variable of every variable, source row of every row)")
            sv, sr = shift_maps
            d.shift_var = self._pin(np.ascontiguousarray(sv, dtype=np.int32)).ctypes.data_as(_ip)
            d.shift_row = self._pin(np.ascontiguousarray(sr, dtype=np.int32)).ctypes.data_as(_ip)
        d.shift_policy = int(shift_policy)
        if stored_start is not None:            # closed loops begin from the solution at this point (e0 = 0), as the device with tz_problem_store_start
            d.start_xbar0 = self._pin(np.ascontiguousarray(stored_start, dtype=np.float64).reshape(qp.n)).ctypes.data_as(_dp)
        self.d = d

    def _pin(self, a):
        self._keep.append(a)
        return a

    @staticmethod
    def max_threads() -> int:
        return lib().tzo_max_threads()

    def solve_batch(self, xbar0, e0, threads: int = 1, want_active: bool = False):
        qp = self.qp
        xbar0 = np.ascontiguousarray(xbar0, dtype=np.float64).reshape(-1, qp.n); e0 = np.ascontiguousarray(e0, dtype=np.float64).reshape(-1, qp.n)
        B = xbar0.shape[0]
        v = np.empty((B, qp.N, qp.m)); xbar = np.empty((B, qp.N + 1, qp.n)); cost = np.empty(B)
        status = np.empty(B, dtype=np.int32); iters = np.empty(B, dtype=np.int32)
        active = np.zeros((B, qp.nc), dtype=np.uint8) if want_active else None
        p = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
        lib().tzo_solve_batch(C.byref(self.d), B, p(xbar0), p(e0), p(v), p(xbar), p(cost), p(status), p(iters), p(active), int(threads))
        return dict(v=v, xbar=xbar, cost=cost, status=status, iters=iters, active=active)

    def simulate_batch(self, x0, noise, A_true, B_true, threads: int = 1, want_iters: bool = False):
        qp = self.qp
        x0 = np.ascontiguousarray(x0, dtype=np.float64).reshape(-1, qp.n); B = x0.shape[0]
        noise = np.ascontiguousarray(noise, dtype=np.float64).reshape(B, -1, qp.n); T = noise.shape[1]
        At = np.ascontiguousarray(A_true, dtype=np.float64); Bt = np.ascontiguousarray(B_true, dtype=np.float64).reshape(qp.n, qp.m)
        xt = np.empty((B, T + 1, qp.n)); ut = np.empty((B, T, qp.m)); cost = np.empty((B, T)); status = np.empty(B, dtype=np.int32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        if want_iters:
            iters = np.zeros((B, T), dtype=np.int32)
            lib().tzo_simulate_batch_iters(C.byref(self.d), B, T, p(x0), p(noise), p(At), p(Bt), p(xt), p(ut), p(cost), p(status), int(threads), p(iters))
            return dict(x=xt, u=ut, cost=cost, status=status, iters=iters)
        lib().tzo_simulate_batch(C.byref(self.d), B, T, p(x0), p(noise), p(At), p(Bt), p(xt), p(ut), p(cost), p(status), int(threads))
        return dict(x=xt, u=ut, cost=cost, status=status)
