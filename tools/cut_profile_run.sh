#!/bin/bash
# GPU box: microseconds per steady-state step up to each checkpoint of tools/cut_profile.py (launch of 30 + K steps minus launch of 30 steps, per step)
# usage: tools/cut_profile_run.sh full.so   (the un-cut dev library for the last line)
run() { TZ_LIB=$1 timeout -k 10 200 python - "$2" <<'PY'
import sys, time, numpy as np
sys.path.insert(0, ".")
import torch
from tests import common
from tzddpc_amd.dist import vertex_noise
ctl, (A, B, zon) = common.gpu_controller("di_n20", warm_shift=3, warm_gain=(0.1, 0.01), mu_factor=1e-5)   # the calibrated settings, given: the cut libraries cannot run the calibration loops
nat = ctl._native; n, m = ctl.qp.n, ctl.qp.m; Bn = 1024
dev = torch.device("cuda", 0)
def loop(K):
    x = torch.from_numpy(np.tile(zon.X0.center, (Bn, 1))).to(dev); xbar = x.clone(); e = torch.zeros_like(x)
    noise = torch.from_numpy(np.ascontiguousarray(vertex_noise(zon.W.compute_vertices(), 0, Bn, K).transpose(1, 0, 2))).to(dev)
    u = torch.zeros((Bn, m), dtype=torch.float64, device=dev); cost = torch.zeros(Bn, dtype=torch.float64, device=dev); st = torch.zeros(Bn, dtype=torch.int32, device=dev)
    At = torch.from_numpy(np.ascontiguousarray(A)).to(dev); Bt = torch.from_numpy(np.ascontiguousarray(B).reshape(n, m)).to(dev)
    best = 1e9
    for rep in range(5):
        x.copy_(torch.from_numpy(np.tile(zon.X0.center, (Bn, 1)))); xbar.copy_(x); e.zero_(); nat.reset_warm(); torch.cuda.synchronize()
        nat.timing_enable(True)
        nat.mpc_run_ptr(Bn, K, x.data_ptr(), xbar.data_ptr(), e.data_ptr(), noise.data_ptr(), At.data_ptr(), Bt.data_ptr(), u.data_ptr(), cost.data_ptr(), st.data_ptr()); nat.sync()
        ms, cnt = nat.timing_get(1); nat.timing_enable(False)
        best = min(best, ms / max(cnt, 1))
    return best
t30, t130 = loop(30), loop(130)
print(f"{sys.argv[1]}: 30 steps {t30:.4f} ms, 130 steps {t130:.4f} ms -> {(t130 - t30) / 100 * 1e3:.3f} us per steady-state step")
PY
}
for k in ${CUTS:-1 2 3 4 5 6 7 8 9 10 11 12 13 14 16}; do run tzddpc_amd/lib/ab/cut$k${SUFFIX}.so "checkpoint $k${SUFFIX}"; done
[ -n "$1" ] && run "$1" "full step"
