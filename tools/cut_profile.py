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
    9: ("      TZ_STAMP(PH_GEMVT);\n      if (!(nrd == nrd))", "before"),                                   # the whole quiet step: every test accepted, epilogue run
}


# second family: every step from the 30th on is made to take exactly one Newton iteration (the acceptance of the starting point is
# switched off), and stops at a checkpoint inside that iteration
FORCE = ("      if (nrd <= pi.tol_res && nrp <= pi.tol_res && mu <= pi.mu_tol) { status = 0; px_in_part = true; break; }",
         "      if (nrd <= pi.tol_res && nrp <= pi.tol_res && mu <= pi.mu_tol && !(fused && step >= 30 && it == 0)) { status = 0; px_in_part = true; break; }")
ITER_CUTS = {
    10: ("    TZ_STAMP(PH_TOP);\n", "before", 1),          # quiet prefix + rejected test + weights staged
    11: ("    TZ_STAMP(PH_FORM);\n", "before", 1),         # + Gram
    12: ("    TZ_FRESH_T();\n    if constexpr (TT) TZ_TT_SOLVE(p, Hq, dinv, r1v, tmpz, dxv);", "before", 1),   # + Cholesky || right-hand side || forward substitution
    13: ("    TZ_STAMP(PH_SOLVE);\n    tz_ell_gemv<MAXR>(p, dxv, pl, rseg_, g_);\n    TZ_STAMP(PH_GEMV);\n    __builtin_amdgcn_s_setprio(TZ_PRIO_ELEM);\n    // step to the boundary", "before", 1),   # + backward substitution
    14: ("    __builtin_amdgcn_s_setprio(TZ_PRIO_ELEM);\n    // step to the boundary", "before", 1),              # + G dx
    15: ("    double sigma = muaff * tz_recip(mu); sigma = sigma * sigma * sigma;", "before", 1),                  # + step lengths (the predictor was not taken as the step)
    16: None,                                                                                                      # whole step with one forced iteration
}


def patched_iter(k):
    s = open(os.path.join(SRC, "tz_ipm.hip.h")).read()
    assert s.count(FORCE[0]) == 1
    s = s.replace(FORCE[0], FORCE[1])
    if ITER_CUTS[k] is None:
        return s
    anchor, where, _ = ITER_CUTS[k]
    assert s.count(anchor) == 1, (k, s.count(anchor))
    code = "    if (fused && step >= 30 && it == 0) { status = 0; it = 0; tz_cut = true; break; }\n"
    s = s.replace(anchor, code + anchor)
    a = "  for (it = 0; it < pk.max_iter && status == 1; ++it) {\n"
    s = s.replace(a, "  bool tz_cut = false;\n" + a)
    b = "  TZ_FRESH_T();\n  work_f = __builtin_amdgcn_readfirstlane(work_f + it"
    s = s.replace(b, "  if (tz_cut) { __syncthreads(); continue; }\n" + b)
    return s


def patched(k):
    if k >= 10:
        return patched_iter(k)
    s = open(os.path.join(SRC, "tz_ipm.hip.h")).read()
    anchor, where = CUTS[k]
    assert s.count(anchor) == 1, (k, s.count(anchor))
    inloop = k in (6, 7)
    if k == 9:                                             # accept whatever the starting point is: no Newton iteration after step 30, the rest of the step as it is
        return s.replace(anchor, "      if (fused && step >= 30) { status = 0; px_in_part = true; break; }\n" + anchor)
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


# what-if variants of a phase (timing only: the results are wrong, the instruction stream is the same), VARIANT=<name> in the environment;
# the libraries are then called cut<k>_<name>.so
_GRAM_FLAG = [   # the step number reaches tz_gram_rows through the (otherwise null) profiling pointer: what-if only in the steps that are timed
    ("tz_gram(p, Hq, Pq, vin, kl, (PROF && t == 0) ? acc_ph : nullptr)", "tz_gram(p, Hq, Pq, vin, kl, (unsigned long long*)(size_t)((fused && step >= 30) ? 1 : 0))"),
    ("  unsigned long long tq0 = pacc ? __builtin_amdgcn_s_memtime() : 0;\n  const int lane = tz_tid() & 63;\n  const int k = lane >> 4, blk",
     "  unsigned long long tq0 = 0; const bool whatif = pacc != nullptr; pacc = nullptr;\n  const int lane = tz_tid() & 63;\n  const int k = lane >> 4, blk"),
]
VARIANTS = {
    # every Gram operand load of a wave hits one of two patch rows: the loads stay (pinned), their addresses are always in L1
    "gram_l1": _GRAM_FLAG + [("    const char* prow = gp + (size_t)kc * rowbytes;\n#pragma unroll\n    for (int J = 0; J < R1; ++J) st.v[J]",
                              "    const char* prow = gp + (size_t)(whatif ? (kc & 1) : kc) * rowbytes;\n#pragma unroll\n    for (int J = 0; J < R1; ++J) st.v[J]")],
    # no operand loads at all: what the matrix instructions, masks and the fold cost by themselves
    "gram_nold": _GRAM_FLAG + [("    for (int J = 0; J < R1; ++J) st.v[J] = tz_ld_pinned((const double*)(prow + (unsigned)(J < Tz ? J : Tz) * 128u));   // tile Tz is zero",
                                "    for (int J = 0; J < R1; ++J) { if (whatif) st.v[J] = st.w + (double)J; else st.v[J] = tz_ld_pinned((const double*)(prow + (unsigned)(J < Tz ? J : Tz) * 128u)); }")],
}


def apply_variant(s):
    v = os.environ.get("VARIANT")
    for a, b in VARIANTS.get(v, []):
        assert s.count(a) == 1, (v, a[:50], s.count(a))
        s = s.replace(a, b)
    return s


def main():
    out = os.path.join(ROOT, "tzddpc_amd", "lib", "ab")
    os.makedirs(out, exist_ok=True)
    flags = ["-mllvm", "-amdgpu-sched-strategy=iterative-maxocc", "-mllvm", "-greedy-regclass-priority-trumps-globalness=1", "-mllvm", "-disable-machine-licm"]
    procs = []
    only = [int(a) for a in sys.argv[1:]]                          # optional: only these checkpoints
    for k in [k for k in list(CUTS) + list(ITER_CUTS) if not only or k in only]:
        base = f"{TMP}{k}{os.environ.get('VARIANT', '')}"
        d = os.path.join(base, "tzddpc_amd", "csrc")                      # the source includes ../../include/tzddpc.h
        shutil.rmtree(base, ignore_errors=True); shutil.copytree(SRC, d); shutil.copytree(os.path.join(ROOT, "include"), os.path.join(base, "include"))
        open(os.path.join(d, "tz_ipm.hip.h"), "w").write(apply_variant(patched(k)))
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-DTZ_ONLY_SMALL"] + flags + \
              ["-I", os.path.join(ROOT, "include"), "-o", os.path.join(out, f"cut{k}" + ("_" + os.environ["VARIANT"] if os.environ.get("VARIANT") else "") + ".so"), os.path.join(d, "tzddpc_hip.hip")]
        procs.append((k, subprocess.Popen(cmd, stderr=subprocess.DEVNULL)))
        if len(procs) % 4 == 0:
            for kk, pr in procs[-4:]:
                assert pr.wait() == 0, kk
    for kk, pr in procs:
        assert pr.wait() == 0, kk
    print("built", sorted(CUTS) + sorted(ITER_CUTS))


if __name__ == "__main__":
    main()
