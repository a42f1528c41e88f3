"""Cumulative cost of a steady-state closed-loop step WITHOUT clocks in the instruction stream: builds copies of the bench-problem kernel
in which every step from the 30th on stops at checkpoint k (state frozen, so every such step runs the same prefix on the same data) and
leaves the libraries in tzddpc_amd/lib/ab/cut<k>.so; time them with tools/cut_profile_run.sh.  The product source is not touched: the
checkpoints are patched into a temporary copy.

    python tools/cut_profile.py          (build container: hipcc cross-compiles)
"""
import os, re, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tzddpc_amd", "csrc")
TMP = "/tmp/cutroot"
CUTS = {
    1: ("  const bool inlaunch = fused && src == 2;\n", "after"),                                           # step top: arguments re-read, policy
    2: ("    TZ_STAMP(PH_TUBE);\n    TZ_COLS(c, nzp) { qv[c]", "before"),                                   # + tube (both stages), shift halves
    3: ("  if (fused && flag[1] != 0) { skip = true;", "before"),                                           # + maps, G x walk, barrier
    4: ("    TZ_STAMP(PH_WARM_B);\n", "before"),                                                            # + owner sums, violation reduction
    5: ("  // scales of the stopping test: parked in LDS", "before"),                                       # + push
    6: ("    TZ_STAMP(PH_T2);\n", "before"),                                                                # + loop top: rp, reduction        (in loop)
    7: ("      TZ_STAMP(PH_GEMVT);\n      if (!(nrd == nrd))", "before"),                                   # + exact dual residual            (in loop)
    8: ("  work_s += 1;\n", "before"),                                                                      # + loop exit, counters
}


def patched(k):
    s = open(os.path.join(SRC, "tz_ipm.hip.h")).read()
    anchor, where = CUTS[k]
    assert s.count(anchor) == 1, (k, s.count(anchor))
    inloop = k in (6, 7)
    code = ("    if (fused && step >= 30) { status = 0; it = 0; tz_cut = true; break; }\n" if inloop
            else "  if (fused && step >= 30) { status = 0; it = 0; __syncthreads(); continue; }\n")
    s = s.replace(anchor, anchor + code if where == "after" else code + anchor)
    if inloop:
        a = "  for (it = 0; it < pk.max_iter && status == 1; ++it) {\n"
        assert s.count(a) == 1
        s = s.replace(a, "  bool tz_cut = false;\n" + a)
        b = "  TZ_FRESH_T();\n  work_f = __builtin_amdgcn_readfirstlane(work_f + it"
        assert s.count(b) == 1
        s = s.replace(b, "  if (tz_cut) { __syncthreads(); continue; }\n" + b)
    return s


def main():
    out = os.path.join(ROOT, "tzddpc_amd", "lib", "ab")
    os.makedirs(out, exist_ok=True)
    flags = ["-mllvm", "-amdgpu-sched-strategy=iterative-maxocc", "-mllvm", "-greedy-regclass-priority-trumps-globalness=1", "-mllvm", "-disable-machine-licm"]
    procs = []
    for k in CUTS:
        base = f"{TMP}{k}"
        d = os.path.join(base, "tzddpc_amd", "csrc")                      # the source includes ../../include/tzddpc.h
        shutil.rmtree(base, ignore_errors=True); shutil.copytree(SRC, d); shutil.copytree(os.path.join(ROOT, "include"), os.path.join(base, "include"))
        open(os.path.join(d, "tz_ipm.hip.h"), "w").write(patched(k))
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-DTZ_ONLY_SMALL"] + flags + \
              ["-I", os.path.join(ROOT, "include"), "-o", os.path.join(out, f"cut{k}.so"), os.path.join(d, "tzddpc_hip.hip")]
        procs.append((k, subprocess.Popen(cmd, stderr=subprocess.DEVNULL)))
        if len(procs) % 4 == 0:
            for kk, pr in procs[-4:]:
                assert pr.wait() == 0, kk
    for kk, pr in procs:
        assert pr.wait() == 0, kk
    print("built", sorted(CUTS))


if __name__ == "__main__":
    main()
