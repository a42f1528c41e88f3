"""ctypes binding of ``libtzddpc_hip.so`` (C-ABI in ``include/tzddpc.h``).

Thin by design: sizes and pointers go in, nothing is computed here.  There is no CPU fallback --
if the shared library is missing or no HIP device is visible the calls raise ``NativeError``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

TZ_ABI_VERSION = 4
TZ_MEM_HOST, TZ_MEM_DEVICE = 0, 1
TZ_SOLVED, TZ_MAX_ITER, TZ_NUMERICAL, TZ_INFEASIBLE = 0, 1, 2, 3

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libtzddpc_hip_prof.so" if os.environ.get("TZ_PROF") == "1" else "libtzddpc_hip.so")
if os.environ.get("TZ_LIB"):                      # diagnostic builds (tools/): explicit library file
    LIB_PATH = os.path.abspath(os.environ["TZ_LIB"])

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


class NativeError(RuntimeError):
    pass


class AffMap(C.Structure):
    _fields_ = [("rows", C.c_int32), ("ptr", _ip), ("col", _ip), ("val", _dp), ("c0", _dp)]


class ProblemDesc(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("n", C.c_int32), ("m", C.c_int32), ("N", C.c_int32),
        ("nz", C.c_int32), ("mi", C.c_int32), ("ntheta", C.c_int32),
        ("P", _dp), ("G", _dp), ("q", AffMap), ("h", AffMap), ("par", AffMap),
        ("par_lo", _dp), ("par_hi", _dp),
        ("cost_scale", C.c_double), ("r0", C.c_double), ("r1", _dp), ("R2", _dp),
        ("Dz", _dp), ("Phi", _dp), ("Gam", _dp),
        ("nc_rows", C.c_int32), ("row_of", _ip), ("act_scale", _dp),
        ("CK", _dp), ("DK", _dp), ("K", _dp), ("pmax", C.c_int32), ("absCKpow", _dp), ("absKCKpow", _dp), ("power", _ip),
        ("max_iter", C.c_int32), ("tol", C.c_double), ("reg", C.c_double), ("step_frac", C.c_double),
        ("shift_var", _ip), ("shift_row", _ip), ("shift_xscale", _dp), ("shift_lscale", _dp),
        ("rec_c0", _dp), ("rec_x0", _dp), ("rec_y", _dp),
        ("plan_flags", C.c_int32),
    ]


# tz_problem_desc.plan_flags (include/tzddpc.h): force the general-size code paths (parity tests hold the paths against each other)
TZ_PLAN_UNFUSED, TZ_PLAN_GENERAL_CHOLESKY, TZ_PLAN_ITEM_GRAM, TZ_PLAN_NO_STAIRCASE, TZ_PLAN_ELL_PRODUCTS = 1, 2, 4, 8, 16


class GenstackDesc(C.Structure):
    _fields_ = [("n", C.c_int32), ("m", C.c_int32), ("N", C.c_int32), ("nseg", C.c_int32), ("seg_ptr", _ip), ("src", _ip),
                ("m0", _dp), ("M", _dp), ("c0", _dp), ("cE", _dp), ("cZ", _dp), ("K", _dp)]


_lib = None


def build_hint() -> str:
    return ("build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(python -c 'import __graft_entry__ as g; g.build()', i.e. hipcc --offload-arch=gfx950 -shared -fPIC tzddpc_amd/csrc/tzddpc_hip.hip)")


def lib():
    """Load the shared library once; declare every symbol of include/tzddpc.h."""
    global _lib
    if _lib is not None:
        return _lib
    try:
        # torch ships its own copy of the HIP runtime: load it first so that the process ends up with ONE runtime
        # (loading ours first leaves torch unable to see the GPU).  torch is plumbing here (tensors, streams, RCCL).
        import torch  # noqa: F401
    except Exception:
        pass
    if not os.path.exists(LIB_PATH):
        raise NativeError(f"{LIB_PATH} not found: the TZDDPC hot path runs only as HIP kernels; {build_hint()}")
    try:
        L = C.CDLL(LIB_PATH)
    except OSError as e:
        raise NativeError(f"cannot load {LIB_PATH}: {e}") from e
    vp = C.c_void_p
    L.tz_abi_version.restype = C.c_int
    L.tz_last_error.restype = C.c_char_p
    L.tz_device_count.argtypes = [C.POINTER(C.c_int)]
    L.tz_problem_create.argtypes = [C.c_int, C.POINTER(ProblemDesc), C.POINTER(vp)]
    L.tz_problem_destroy.argtypes = [vp]
    L.tz_problem_set_stream.argtypes = [vp, vp]
    L.tz_problem_sync.argtypes = [vp]
    L.tz_solve_batch.argtypes = [vp, C.c_int32, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int]
    L.tz_simulate_batch.argtypes = [vp, C.c_int32, C.c_int32, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int]
    L.tz_mpc_step.argtypes = [vp, C.c_int32, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.tz_mpc_run.argtypes = [vp, C.c_int32, C.c_int32] + [vp] * 9
    L.tz_identify_batch.argtypes = [C.c_int, C.c_int32, C.c_int32, C.c_int32, C.c_int32] + [vp] * 4 + [C.c_int32] + [vp] * 5 + [C.c_int]
    L.tz_identify_batch.restype = C.c_int
    L.tz_specrad_batch.argtypes = [C.c_int, C.c_int32, C.c_int32, C.c_int32, vp, vp, vp, vp, vp]
    L.tz_specrad_batch.restype = C.c_int
    L.tz_adversary_batch.argtypes = [C.c_int, C.c_int32, C.c_int32, C.c_int32, vp, vp, vp, C.c_int32, vp, vp, vp]
    L.tz_adversary_batch.restype = C.c_int
    L.tz_genstack_create.argtypes = [C.c_int, C.POINTER(GenstackDesc), C.POINTER(vp)]
    L.tz_genstack_destroy.argtypes = [vp]
    L.tz_genstack_intervals.argtypes = [vp, C.c_int32, vp, vp, vp, vp, vp, C.POINTER(C.c_double), C.c_int]
    L.tz_genstack_values.argtypes = [vp, C.c_int32, C.c_int32, vp, vp, vp, C.c_int]
    L.tz_genstack_info.argtypes = [vp] + [C.POINTER(C.c_int64)] * 3
    for nm in ("tz_problem_attach_tube_stack", "tz_genstack_create", "tz_genstack_destroy", "tz_genstack_intervals", "tz_genstack_values", "tz_genstack_info"):
        getattr(L, nm).restype = C.c_int
    L.tz_problem_reset_warm.argtypes = [vp]
    L.tz_problem_store_start.argtypes = [vp, vp, vp]
    L.tz_problem_store_start.restype = C.c_int
    L.tz_problem_attach_tube_stack.argtypes = [vp, vp]
    L.tz_problem_attach_tube_stack.restype = C.c_int
    L.tz_problem_set_stopping.argtypes = [vp, C.c_double, C.c_double]
    L.tz_problem_set_stopping.restype = C.c_int
    L.tz_problem_set_warm_quiet.argtypes = [vp, C.c_int32]
    L.tz_problem_set_warm_quiet.restype = C.c_int
    L.tz_problem_set_warm_push.argtypes = [vp, C.c_double, C.c_double, C.c_double]
    L.tz_problem_set_warm_push.restype = C.c_int
    L.tz_problem_set_warm_shift.argtypes = [vp, C.c_int32]
    L.tz_timing_enable.argtypes = [vp, C.c_int]
    L.tz_timing_get.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    L.tz_ipm_plan_info.argtypes = [vp] + [C.POINTER(C.c_int64)] * 5
    L.tz_ipm_work_get.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.tz_problem_plan_get.argtypes = [vp] + [C.POINTER(C.c_int32)] * 5
    L.tz_problem_plan_get.restype = C.c_int
    L.tz_debug_fetch.argtypes = [vp, C.c_int32, C.c_int, vp, C.c_int32]
    for name in ("tz_device_count", "tz_problem_create", "tz_problem_destroy", "tz_problem_set_stream", "tz_problem_sync",
                 "tz_solve_batch", "tz_simulate_batch", "tz_mpc_step", "tz_mpc_run", "tz_timing_enable", "tz_timing_get",
                 "tz_ipm_plan_info", "tz_ipm_work_get", "tz_debug_fetch", "tz_problem_reset_warm", "tz_problem_set_warm_shift"):
        getattr(L, name).restype = C.c_int
    if L.tz_abi_version() != TZ_ABI_VERSION:
        raise NativeError(f"ABI mismatch: library {L.tz_abi_version()} vs binding {TZ_ABI_VERSION}")
    _lib = L
    return L


EXPORTED_SYMBOLS = ("tz_abi_version", "tz_last_error", "tz_device_count", "tz_problem_create", "tz_problem_destroy", "tz_problem_plan_get", "tz_problem_store_start",
                    "tz_problem_set_stream", "tz_problem_sync", "tz_solve_batch", "tz_simulate_batch", "tz_mpc_step", "tz_mpc_run",
                    "tz_timing_enable", "tz_timing_get", "tz_ipm_plan_info", "tz_ipm_work_get", "tz_debug_fetch",
                    "tz_problem_set_warm_shift", "tz_problem_set_stopping", "tz_problem_set_warm_quiet", "tz_problem_set_warm_push", "tz_problem_reset_warm", "tz_identify_batch", "tz_specrad_batch", "tz_adversary_batch",
                    "tz_problem_attach_tube_stack", "tz_genstack_create", "tz_genstack_destroy", "tz_genstack_intervals", "tz_genstack_values", "tz_genstack_info")


def check(rc: int, what: str):
    if rc < 0:
        raise NativeError(f"{what} failed ({rc}): {lib().tz_last_error().decode(errors='replace')}")
    return rc


def device_count() -> int:
    n = C.c_int(0)
    check(lib().tz_device_count(C.byref(n)), "tz_device_count")
    return n.value


def identify_batch(device: int, u: np.ndarray, x: np.ndarray, w_center: np.ndarray, K: Optional[np.ndarray] = None):
    """K0 (``tz_identify_batch``): u (B, T, m), x (B, T, n) -> dict(C (B, n, n+m), s (B, n+m)[, sK (B, n), CK (B, n, n)], status)."""
    u = np.ascontiguousarray(u, dtype=np.float64); x = np.ascontiguousarray(x, dtype=np.float64)
    if u.ndim == 2:
        u = u[None]; x = x[None]
    B, T, m = u.shape
    n = x.shape[2]
    wc = np.ascontiguousarray(w_center, dtype=np.float64).reshape(n)
    Cc = np.empty((B, n, n + m)); s = np.empty((B, n + m)); status = np.empty(B, dtype=np.int32)
    sK = CK = Kc = None
    shared = 0
    if K is not None:
        Kc = np.ascontiguousarray(K, dtype=np.float64)
        shared = 1 if Kc.ndim == 2 else 0
        Kc = Kc.reshape((m, n) if shared else (B, m, n))
        sK = np.empty((B, n)); CK = np.empty((B, n, n))
    vp = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
    check(lib().tz_identify_batch(int(device), B, T, n, m, vp(u), vp(x), vp(wc), vp(Kc), shared, vp(Cc), vp(s), vp(sK), vp(CK), vp(status),
                                  TZ_MEM_HOST), "tz_identify_batch")
    out = dict(C=Cc, s=s, status=status)
    if K is not None:
        out.update(sK=sK, CK=CK)
    return out


def specrad_batch(device: int, M0: np.ndarray, H: np.ndarray, beta: np.ndarray):
    """``tz_specrad_batch``: spectral radius of M0 + sum_i beta[s, i] H[i] for every row s of beta -> (rho (S,), status (S,))."""
    M0 = np.ascontiguousarray(M0, dtype=np.float64); n = M0.shape[0]
    H = np.ascontiguousarray(H, dtype=np.float64).reshape(-1, n, n)
    beta = np.ascontiguousarray(beta, dtype=np.float64).reshape(-1, H.shape[0])
    S = beta.shape[0]
    rho = np.empty(S); status = np.empty(S, dtype=np.int32)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    check(lib().tz_specrad_batch(int(device), S, n, H.shape[0], vp(M0), vp(H), vp(beta), vp(rho), vp(status)), "tz_specrad_batch")
    return rho, status


def adversary_batch(device: int, M0: np.ndarray, H: np.ndarray, beta0: np.ndarray, max_iter: int = 100):
    """``tz_adversary_batch``: CCP ascent of ||M0 + sum_i beta_i H_i||_F over the box from every row of beta0
    -> (beta (S, ngen) fixed points, fro (S,), steps (S,))."""
    M0 = np.ascontiguousarray(M0, dtype=np.float64); n = M0.shape[0]
    H = np.ascontiguousarray(H, dtype=np.float64).reshape(-1, n, n)
    beta0 = np.ascontiguousarray(beta0, dtype=np.float64).reshape(-1, H.shape[0])
    S = beta0.shape[0]
    beta = np.empty_like(beta0); fro = np.empty(S); steps = np.empty(S, dtype=np.int32)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    check(lib().tz_adversary_batch(int(device), S, n, H.shape[0], vp(M0), vp(H), vp(beta0), int(max_iter), vp(beta), vp(fro), vp(steps)),
          "tz_adversary_batch")
    return beta, fro, steps


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _ptr(a, typ):
    return a.ctypes.data_as(typ)


def _csr(M: np.ndarray, c0: np.ndarray, keep: list):
    """dense (rows x ntheta) -> AffMap; `keep` pins the numpy buffers for the lifetime of the desc."""
    M = np.atleast_2d(_f64(M))
    rows = M.shape[0] if M.size else len(c0)
    ptr = np.zeros(rows + 1, dtype=np.int32)
    cols, vals = [], []
    for r in range(rows):
        nzc = np.nonzero(M[r])[0] if M.size else np.zeros(0, dtype=int)
        cols.append(nzc); vals.append(M[r, nzc] if M.size else np.zeros(0))
        ptr[r + 1] = ptr[r] + len(nzc)
    col = _i32(np.concatenate(cols) if cols else np.zeros(0))
    val = _f64(np.concatenate(vals) if vals else np.zeros(0))
    if col.size == 0:
        col = np.zeros(1, dtype=np.int32); val = np.zeros(1)
    c0 = _f64(c0) if len(c0) else np.zeros(1)
    keep += [ptr, col, val, c0]
    return AffMap(rows, _ptr(ptr, _ip), _ptr(col, _ip), _ptr(val, _dp), _ptr(c0, _dp))


class Problem:
    """Owner of one ``tz_problem`` handle."""

    def __init__(self, device: int, *, n, m, N, P, G, q0, Qt, h0, Ht, par0, Part, par_lo, par_hi, cost_scale, r0, r1, R2,
                 Dz, Phi, Gam, nc_rows, row_of, act_scale, CK, DK, K, pmax, absCKpow, absKCKpow, power,
                 max_iter=40, tol=1e-10, reg=1e-12, step_frac=0.99999,
                 shift_var=None, shift_row=None, shift_xscale=None, shift_lscale=None, rec_c0=None, rec_x0=None, rec_y=None, plan_flags=0):
        L = lib()
        keep = []
        d = ProblemDesc()
        d.abi_version = TZ_ABI_VERSION
        d.n, d.m, d.N = int(n), int(m), int(N)
        P = _f64(P); G = _f64(G)
        d.nz, d.mi = P.shape[0], G.shape[0]
        d.ntheta = 2 * n + N * (2 * n + m)
        arrs = dict(P=P, G=G, r1=_f64(r1), R2=_f64(R2), Dz=_f64(Dz), Phi=_f64(Phi), Gam=_f64(Gam), act_scale=_f64(act_scale), CK=_f64(CK), DK=_f64(DK),
                    K=_f64(K), absCKpow=_f64(absCKpow), absKCKpow=_f64(absKCKpow))
        for k, a in arrs.items():
            setattr(d, k, _ptr(a, _dp)); keep.append(a)
        d.q = _csr(Qt, q0, keep); d.h = _csr(Ht, h0, keep); d.par = _csr(Part, par0, keep)
        plo = _f64(par_lo) if len(par_lo) else np.zeros(1); phi = _f64(par_hi) if len(par_hi) else np.zeros(1)
        keep += [plo, phi]
        d.par_lo = _ptr(plo, _dp); d.par_hi = _ptr(phi, _dp)
        d.cost_scale = float(cost_scale); d.r0 = float(r0)
        ro = _i32(row_of); pw = _i32(power); keep += [ro, pw]
        d.nc_rows = int(nc_rows); d.row_of = _ptr(ro, _ip)
        d.pmax = int(pmax); d.power = _ptr(pw, _ip)
        d.max_iter = int(max_iter); d.tol = float(tol); d.reg = float(reg); d.step_frac = float(step_frac)
        d.plan_flags = int(plan_flags)
        if shift_var is not None:
            sv, sr, xs, ls = _i32(shift_var), _i32(shift_row), _f64(shift_xscale), _f64(shift_lscale)
            keep += [sv, sr, xs, ls]
            d.shift_var = _ptr(sv, _ip); d.shift_row = _ptr(sr, _ip); d.shift_xscale = _ptr(xs, _dp); d.shift_lscale = _ptr(ls, _dp)
        if rec_y is not None:
            rc, rx, ry = _f64(rec_c0).reshape(N * m), _f64(rec_x0).reshape(N * m, n), _f64(rec_y).reshape(N * m, d.nz)
            keep += [rc, rx, ry]
            d.rec_c0 = _ptr(rc, _dp); d.rec_x0 = _ptr(rx, _dp); d.rec_y = _ptr(ry, _dp)
        self.n, self.m, self.N, self.nz, self.mi, self.nc_rows, self.ntheta = int(n), int(m), int(N), d.nz, d.mi, int(nc_rows), d.ntheta
        # structure-aware algorithmic work of one interior-point factorisation + its two solves (what bench.py's roofline
        # counts): sparse outer products of the rows of G, Cholesky, four G / G' products, two triangular solve pairs, P x
        Gn = np.asarray(G).reshape(d.mi, d.nz); nnz_r = (Gn != 0).sum(axis=1).astype(np.float64)
        self.alg_flops = dict(gram=float((nnz_r * (nnz_r + 1)).sum()), cholesky=float(d.nz) ** 3 / 3.0,
                              g_products=8.0 * float(nnz_r.sum()), solves=8.0 * float(d.nz) ** 2,
                              p_products=2.0 * float((np.asarray(P) != 0).sum()))
        self.alg_flops["per_factorization"] = float(sum(self.alg_flops.values()))
        # fixed part of one closed-loop step (F_tube = sum_k 2 n^2 (k n + n) + 2 m n (k n)), the three affine maps
        # over theta, recovery of xbar[1] and the plant / error update
        nmap = float(np.count_nonzero(Qt) + np.count_nonzero(Ht) + np.count_nonzero(Part))
        f_tube = float(sum(2 * n * n * (k * n + n) + 2 * m * n * (k * n) for k in range(int(N))))
        self.alg_flops["per_step_fixed"] = f_tube + 2.0 * nmap + 2.0 * n * (n + N * m) + 2.0 * n * (n + m) + 2.0 * m * n
        self.alg_flops["dense_per_factorization"] = float(d.mi * d.nz * (d.nz + 1) + d.nz ** 3 / 3.0 + 8.0 * d.mi * d.nz + 10.0 * d.nz ** 2)
        h = C.c_void_p()
        check(L.tz_problem_create(int(device), C.byref(d), C.byref(h)), "tz_problem_create")
        self._h = h
        self.device = int(device)
        self.stored_start = None

    def close(self):
        if getattr(self, "_h", None):
            lib().tz_problem_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- host-pointer entry points -------------------------------------------------------------
    def solve_batch(self, xbar0: np.ndarray, e0: np.ndarray, want_active: bool = False):
        xbar0 = _f64(xbar0).reshape(-1, self.n); e0 = _f64(e0).reshape(-1, self.n)
        B = xbar0.shape[0]
        if e0.shape[0] != B:
            raise ValueError("xbar0 and e0 must have the same batch size")
        v = np.empty((B, self.N, self.m)); xbar = np.empty((B, self.N + 1, self.n)); cost = np.empty(B)
        status = np.empty(B, dtype=np.int32); iters = np.empty(B, dtype=np.int32)
        active = np.zeros((B, self.nc_rows), dtype=np.uint8) if want_active else None
        vp = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
        check(lib().tz_solve_batch(self._h, B, vp(xbar0), vp(e0), vp(v), vp(xbar), vp(cost), vp(status), vp(iters), vp(active),
                                   TZ_MEM_HOST), "tz_solve_batch")
        return v, xbar, cost, status, iters, active

    def simulate_batch(self, x0, noise, A_true, B_true):
        x0 = _f64(x0).reshape(-1, self.n); B = x0.shape[0]
        noise = _f64(noise).reshape(B, -1, self.n); T = noise.shape[1]
        A_true = _f64(A_true); B_true = _f64(B_true).reshape(self.n, self.m)
        xt = np.empty((B, T + 1, self.n)); ut = np.empty((B, T, self.m)); cost = np.empty((B, T)); status = np.empty(B, dtype=np.int32)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        check(lib().tz_simulate_batch(self._h, B, T, vp(x0), vp(noise), vp(A_true), vp(B_true), vp(xt), vp(ut), vp(cost), vp(status),
                                      TZ_MEM_HOST), "tz_simulate_batch")
        return xt, ut, cost, status

    # ---- device-pointer entry points (torch tensors: pass .data_ptr()) --------------------------
    def solve_batch_ptr(self, B, xbar0, e0, v, xbar, cost, status, iters=None, active=None):
        check(lib().tz_solve_batch(self._h, int(B), xbar0, e0, v, xbar, cost, status, iters, active, TZ_MEM_DEVICE), "tz_solve_batch")

    def mpc_step_ptr(self, B, x, xbar, e, w, A_true, B_true, u_out, cost, status):
        check(lib().tz_mpc_step(self._h, int(B), x, xbar, e, w, A_true, B_true, u_out, cost, status), "tz_mpc_step")

    def mpc_run_ptr(self, B, K, x, xbar, e, w, A_true, B_true, u_out, cost, status):
        check(lib().tz_mpc_run(self._h, int(B), int(K), x, xbar, e, w, A_true, B_true, u_out, cost, status), "tz_mpc_run")

    def simulate_batch_ptr(self, B, T, x0, noise, A_true, B_true, x_traj, u_traj, cost, status):
        check(lib().tz_simulate_batch(self._h, int(B), int(T), x0, noise, A_true, B_true, x_traj, u_traj, cost, status,
                                      TZ_MEM_DEVICE), "tz_simulate_batch")

    def set_stream(self, stream_ptr: Optional[int]):
        check(lib().tz_problem_set_stream(self._h, stream_ptr), "tz_problem_set_stream")

    def sync(self):
        check(lib().tz_problem_sync(self._h), "tz_problem_sync")

    def reset_warm(self):
        """The next closed-loop call starts a new batch of trajectories: no warm start from what the handle solved before."""
        check(lib().tz_problem_reset_warm(self._h), "tz_problem_reset_warm")

    def attach_tube_stack(self, stack: "GenStack"):
        """Literal problems: theta's tube block comes from the device evaluation of `stack` (``tz_problem_attach_tube_stack``)."""
        check(lib().tz_problem_attach_tube_stack(self._h, stack._h if stack is not None else None), "tz_problem_attach_tube_stack")
        self._tube_stack = stack                      # keeps it alive as long as the problem

    def set_stopping(self, res_factor: float = 100.0, mu_factor: float = 1e-3):
        check(lib().tz_problem_set_stopping(self._h, float(res_factor), float(mu_factor)), "tz_problem_set_stopping")
        self.stopping = (float(res_factor), float(mu_factor))          # last values handed to the library (read back by tests)

    def set_warm_quiet(self, quiet_steps: int = 16):
        check(lib().tz_problem_set_warm_quiet(self._h, int(quiet_steps)), "tz_problem_set_warm_quiet")

    def set_warm_push(self, floor: float = 1e-8, gain: float = 1.0, cap: float = 1e300):
        check(lib().tz_problem_set_warm_push(self._h, float(floor), float(gain), float(min(cap, 1e300))), "tz_problem_set_warm_push")
        self.warm_push = (float(floor), float(gain), float(cap))

    def store_start(self, xbar0=None, e0=None):
        """Solve once at (xbar0, e0) and start fresh closed loops from that solution (``tz_problem_store_start``); None forgets it."""
        if xbar0 is None:
            check(lib().tz_problem_store_start(self._h, None, None), "tz_problem_store_start")
            self.stored_start = None
            return
        a = _f64(xbar0).reshape(self.n); b = _f64(np.zeros(self.n) if e0 is None else e0).reshape(self.n)
        check(lib().tz_problem_store_start(self._h, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p)), "tz_problem_store_start")
        self.stored_start = (a.copy(), b.copy())

    def set_warm_shift(self, policy: int):
        check(lib().tz_problem_set_warm_shift(self._h, int(policy)), "tz_problem_set_warm_shift")

    def timing_enable(self, on: bool = True):
        check(lib().tz_timing_enable(self._h, int(on)), "tz_timing_enable")

    def timing_get(self, kernel: int):
        ms = C.c_double(0); cnt = C.c_int64(0)
        check(lib().tz_timing_get(self._h, kernel, C.byref(ms), C.byref(cnt)), "tz_timing_get")
        return ms.value, cnt.value

    def work_get(self):
        a, b, c = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        check(lib().tz_ipm_work_get(self._h, C.byref(a), C.byref(b), C.byref(c)), "tz_ipm_work_get")
        return dict(factorizations=a.value, trajectory_solves=b.value, max_factorizations_one_trajectory=c.value)

    def plan_info(self):
        a, b, i, c, d = (C.c_int64(0) for _ in range(5))
        check(lib().tz_ipm_plan_info(self._h, C.byref(a), C.byref(b), C.byref(i), C.byref(c), C.byref(d)), "tz_ipm_plan_info")
        f = [C.c_int32(0) for _ in range(5)]
        check(lib().tz_problem_plan_get(self._h, *[C.byref(v) for v in f]), "tz_problem_plan_get")
        return dict(mfma_gram_per_iter=a.value, mfma_chol_per_iter=b.value, mfma_issued_per_iter=i.value,
                    lds_bytes=c.value, patch_bytes=d.value, fused=bool(f[0].value), chol1=bool(f[1].value), ksplit=bool(f[2].value),
                    staircase=bool(f[3].value), toeplitz=bool(f[4].value))

    def last_iterations(self, B: int) -> np.ndarray:
        """Interior-point iterations of each trajectory in the last launch (diagnostic)."""
        out = np.empty(int(B))
        n = check(lib().tz_debug_fetch(self._h, 0, 7, out.ctypes.data_as(C.c_void_p), int(B)), "tz_debug_fetch")
        return out[:n].astype(np.int64)

    def debug_fetch(self, b: int, what: int) -> np.ndarray:
        cap = max(self.nz, self.mi, self.ntheta, 64)
        out = np.empty(cap)
        n = check(lib().tz_debug_fetch(self._h, int(b), int(what), out.ctypes.data_as(C.c_void_p), cap), "tz_debug_fetch")
        return out[:n].copy()


class GenStack:
    """Owner of one ``tz_genstack`` handle: the literal stacked-generator tubes of a problem on the device (kernel K1g)."""

    def __init__(self, device: int, st):
        d = GenstackDesc()
        keep = []
        d.n, d.m, d.N, d.nseg = int(st.n), int(st.m), int(st.N), int(st.nseg)
        sp = _i32(st.seg_ptr); sr = _i32(st.src if st.src.size else np.zeros(1)); keep += [sp, sr]
        d.seg_ptr = _ptr(sp, _ip); d.src = _ptr(sr, _ip)
        for name in ("m0", "M", "c0", "cE", "K"):
            a = _f64(getattr(st, name))
            if a.size == 0:
                a = np.zeros(1)
            keep.append(a); setattr(d, name, _ptr(a, _dp))
        if np.any(st.cZ):
            a = _f64(st.cZ); keep.append(a); d.cZ = _ptr(a, _dp)
        self.n, self.m, self.N, self.nseg = d.n, d.m, d.N, d.nseg
        self.num_generators = np.diff(np.asarray(st.seg_ptr)).astype(np.int64)
        h = C.c_void_p()
        check(lib().tz_genstack_create(int(device), C.byref(d), C.byref(h)), "tz_genstack_create")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            lib().tz_genstack_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _inputs(self, e0, zeta):
        e0 = _f64(e0).reshape(-1, self.n); B = e0.shape[0]
        zeta = _f64(zeta).reshape(B, self.N, self.n + self.m)
        return e0, zeta, B

    def intervals(self, e0, zeta, want_ms: bool = False):
        """(centre (B, nseg, n), rad_x (B, nseg, n), rad_u (B, nseg, m)) of Ze[k] and K Ze[k] for every trajectory."""
        e0, zeta, B = self._inputs(e0, zeta)
        c = np.empty((B, self.nseg, self.n)); rx = np.empty((B, self.nseg, self.n)); ru = np.empty((B, self.nseg, self.m))
        ms = C.c_double(0.0)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        check(lib().tz_genstack_intervals(self._h, B, vp(e0), vp(zeta), vp(c), vp(rx), vp(ru), C.byref(ms) if want_ms else None, TZ_MEM_HOST),
              "tz_genstack_intervals")
        return (c, rx, ru, ms.value) if want_ms else (c, rx, ru)

    def intervals_ptr(self, B, e0, zeta, centre, rad_x, rad_u):
        """Device pointers; returns the HIP-event time of the streaming kernel in ms."""
        ms = C.c_double(0.0)
        check(lib().tz_genstack_intervals(self._h, int(B), e0, zeta, centre, rad_x, rad_u, C.byref(ms), TZ_MEM_DEVICE), "tz_genstack_intervals")
        return ms.value

    def values(self, seg: int, e0, zeta):
        """[centre | generators] of Ze[seg] in the reference's column order: (B, n, 1 + Gamma_seg)."""
        e0, zeta, B = self._inputs(e0, zeta)
        Z = np.empty((B, self.n, 1 + int(self.num_generators[seg])))
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        check(lib().tz_genstack_values(self._h, int(seg), B, vp(e0), vp(zeta), vp(Z), TZ_MEM_HOST), "tz_genstack_values")
        return Z

    def info(self):
        a, b, c = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        check(lib().tz_genstack_info(self._h, C.byref(a), C.byref(b), C.byref(c)), "tz_genstack_info")
        return dict(generators=a.value, stack_bytes=b.value, chunks=c.value)
