"""-m "not gpu": the plain-C oracle (oracle/c/tz_oracle.c -- the full-size checker of every configuration and the CPU baseline)
built with AddressSanitizer + UndefinedBehaviorSanitizer and driven through the paths the parity tests use: stateless solves with
the active set, warm-started closed loops with and without the horizon shift, equality-free and LP-type problems, OpenMP teams.
The GPU pool offers no sanitizers: this is where out-of-bounds accesses of the checker would show."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_DRIVER = r'''
import sys
sys.path.insert(0, sys.argv[1])
import numpy as np
from tests import common
from oracle.c_oracle import COracle
from tzddpc_amd.builder import horizon_shift
from tzddpc_amd.dist import vertex_noise
for case, pol, T, Bn in (("di_n5", 3, 12, 5), ("di_n20", 3, 8, 4), ("pulley_n10", 0, 6, 3), ("di_n20_k1", 0, 4, 3), ("di2in_n10", 3, 6, 3)):
    ctl, qp, (A, B, zon) = common.identified_qp(case)
    co = COracle(qp, shift_policy=pol, shift_maps=horizon_shift(qp) if pol else None, warm_gain=0.1, warm_cap=0.01, mu_factor=1e-4)
    x0, e0 = common.sample_params(zon, qp.n, Bn)
    out = co.solve_batch(x0, e0, threads=2, want_active=True)
    assert (out["status"] == 0).all(), (case, out["status"])
    ref = common.oracle_solution(qp, x0[1], e0[1])
    assert abs(out["cost"][1] - ref["cost"]) <= 1e-7 * (1 + abs(ref["cost"])), case
    sim = co.simulate_batch(np.tile(zon.X0.center, (Bn, 1)), vertex_noise(zon.W.compute_vertices(), 0, Bn, T), A, B, threads=3, want_iters=True)
    assert (sim["status"] == 0).all(), case
    bad = co.solve_batch(x0 + 50.0, e0, threads=1)                      # far outside X: parameter rows violated / infeasible, the failure paths
    assert (bad["status"] != 0).all(), case
maps = open("/proc/self/maps").read()
assert "libtz_oracle_asan" in maps and "libasan" in maps            # the instrumented build is what ran
print("sanitized oracle ok")
'''


@pytest.mark.timeout(900)
def test_c_oracle_under_asan_and_ubsan(tmp_path):
    gcc = os.environ.get("CC", "gcc")
    libasan = subprocess.run([gcc, "-print-file-name=libasan.so"], stdout=subprocess.PIPE, text=True).stdout.strip()
    if not libasan or not os.path.exists(libasan):
        pytest.skip("no libasan in this toolchain")
    out = str(tmp_path / "libtz_oracle_asan.so")
    subprocess.check_call([gcc, "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-fopenmp", "-fPIC", "-std=c11", "-Wall", "-shared", "-o", out, os.path.join(ROOT, "oracle", "c", "tz_oracle.c"), "-lm"])
    env = dict(os.environ, LD_PRELOAD=libasan, TZ_ORACLE_LIB=out, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               OMP_NUM_THREADS="3")
    drv = tmp_path / "drive.py"
    drv.write_text(_DRIVER)
    p = subprocess.run([sys.executable, str(drv), ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=800)
    assert p.returncode == 0 and "sanitized oracle ok" in p.stdout, (p.stdout[-1500:], p.stderr[-3000:])
    assert "runtime error" not in p.stderr and "AddressSanitizer" not in p.stderr, p.stderr[-3000:]
