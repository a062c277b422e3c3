# thermal_3D.jl -- the reference's examples/thermal_conduction/3D_Script.jl on the MI355X backend, UNCHANGED.
#
# STATUS: WRITTEN, NOT EXECUTED (no Julia in the build image, SURVEY.md F2); every `ccall` it reaches is checked against include/metafem_mi355x.h by
# tests/test_julia_binding.py.
#
# What "existing examples/*.jl run unchanged" (BASELINE.json north_star) means in practice: the script is NOT copied or edited -- it is included from
# the MetaFEM.jl checkout as it ships, after two lines that switch the array backend and install the four seams (julia/MI355X.jl: install!):
#
#     S0  array type            misc/04_GPU_Utils.jl:1-38         CuArray -> ROCArray (FEM_zeros / FEM_convert / ...)
#     S1  linear solver         solver/01_Types.jl:166            the script's own line `fem_domain.linear_solver = x -> iterative_Solve!(x; Sv_func! = idrs!,
#                                                                 maxiter = 2000, max_pass = 10, s = 8)` (3D_Script.jl:49) now dispatches to mfem_solve
#     S3  element operators     solver/06_FEM_Kernel.jl:1,28,65   called by the updaters `compile_Updater_GPU` generates (3D_Script.jl:40) -> mfem_op_var / _kval / _res
#     S2  (optional fast path)  solver/01_Types.jl:164-165        constant-coefficient forms on bricks: MI355X.install_thermal_fastpath! replaces the two generated
#                                                                 closures by the fused kernels -- not used below: this mesh is an unstructured tet-10 mesh
#
# The lines of the reference script that matter to the backend, for orientation (nothing to change in any of them):
#     :38  mesh_Classical(...; itp_type = :Serendipity, itp_order = 2, itg_order = 5, ...)   tables on the host, arrays through FEM_convert (S0)
#     :40  compile_Updater_GPU(...)                                                          emits calls of the S3 operators
#     :42-44 update_Mesh / :45 assemble_Global_Variables!                                    geometry tables, sparse pattern (INTEGRATION.md: the two C calls)
#     :49  fem_domain.linear_solver = x -> iterative_Solve!(x; Sv_func! = idrs!, ...)        S1
#     :58  update_OneStep!(...)                                                              solver/04_Time_Domain.jl:59-80, unchanged Julia
using MetaFEM

include(joinpath(@__DIR__, "..", "MI355X.jl"))
MI355X.install!(MetaFEM)

include(joinpath(pkgdir(MetaFEM), "examples", "thermal_conduction", "3D_Script.jl"))
