"""CPU oracle for the MetaFEM.jl assembly-and-solve hot path.

TEST INFRASTRUCTURE ONLY.  This package is a CPU restatement (numpy/scipy, plus
the C/OpenMP file under ``oracle/c``) of the reference algorithm
(jxx2/MetaFEM.jl v0.1.4, 100 % Julia on CUDA.jl).  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it; the product package ``metafem.jl_amd`` never does and fails loudly
when its HIP library is missing.

The reference cannot be executed in the build container (no Julia, and it has
no CPU kernel path at all), so every function here cites the
reference ``file:line`` it restates.  Pin: the restatement reproduces the
reference's own committed output ``examples/thermal_conduction/
2D_Ceramic_Strip.vtk`` (quad-8 serendipity, Nitsche/penalty Dirichlet, convective
+ radiative boundary, Newton) when matched by coordinates -- see
``tests/test_oracle_golden.py`` and ``tests/golden/make_golden.py``.
hex-8 / hex-27 Lagrange elements have no reference-produced output (SURVEY.md
F12); they are pinned by the shared code path plus patch / manufactured-solution
tests.
"""
