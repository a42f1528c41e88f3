"""CPU oracle for the TZDDPC hot path  --  TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement of the algorithm behind
``TZDDPC.build_problem`` / ``TZDDPC.solve`` (reference ``tzddpc/tzddpc.py:132-241, 357-377``).
It exists to *check* the HIP product in ``tzddpc_amd/``; nothing in ``tzddpc_amd/`` may import,
call, link or execute anything under ``oracle/``.  Only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` use it.

PARITY UNPINNED.  The reference has no tests, golden vectors or fixtures for this path and its
arithmetic lives in un-vendored, un-pinned third-party packages (``pyzonotope``,
``pydatadrivenreachability``, ``cvxpy`` + conic solver; reference ``setup.py:12``) that are not
installed in this image, so the reference cannot be imported or run here (plain
``ModuleNotFoundError: No module named 'cvxpy'`` at ``tzddpc/tzddpc.py:2``; nothing was denied).
The oracle is therefore pinned only by
  * hand-computed known answers for the zonotope algebra (``tests/test_oracle_zonolite.py``),
  * the identity literal-generator-stacking == collapsed form (``tests/test_oracle_collapse.py``),
  * KKT certificates (<=1e-8) stored with every golden solution, so no solver is trusted,
  * two independent solvers (own interior point, scipy HiGHS) agreeing on LP-type cases.

Modules
  zonolite    literal zonotope / matrix-zonotope / interval algebra (CORA semantics that
              ``pyzonotope`` ports), incl. decision-variable-affine zonotopes (CVXZonotope).
  literal     line-by-line restatement of ``build_problem`` / ``build_problem_simplified``
              with literal generator stacking (exponential in the horizon -> small N only).
  collapsed   the same problem with exactly-aggregated generators (any N).
  qp_ipm      dense primal-dual interior-point QP solver + KKT certificate.
  harness     data generation / closed loop restated from ``examples/``.
  c/          plain-C restatement of the per-step numeric path (CPU baseline + kernel checker).
"""
