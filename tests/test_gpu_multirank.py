"""The real multi-rank solver path, executed: 2 and 3 ranks on ONE GPU.

tests/multirank_worker.py is started `world` times as fresh child processes on cuda:0.  Each rank owns a slab of a structured
problem, attaches the host-callback communicator (mfem_comm_create_host; gloo moves the staged data, because RCCL refuses
ranks that share a device) and runs the product's multi-rank code: halo exchange (ghost planes bitwise), reverse halo
(global column norms), |diag| with halo (bitwise), device-scalar all-reduce, and mfem_solve with CG (classic recurrence: same
iteration count as one rank, <= 1e-12; single-reduction form: <= 1e-10), BiCGStab(2) and IDR(8) (same shadow vectors: <= 1e-8),
with the halo exchange overlapped with the interior rows and blocking.  Ghost entries are NaN between halo begin and end, so
an interior kernel that read them early would poison the result.  The RCCL transport itself needs >= 2 GPUs (driver's scaling
run); everything above it -- row split, zones, fold + all-reduce groups, ghost offsets -- is what runs here.
"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_ranks(case, world, timeout=900, transport=None):
    """Start the ranks as fresh processes; their output goes to temporary FILES (a rank that fills a 64 KB pipe while rank 0 waits
    for it in a collective would hang the test until its timeout), one deadline for the whole group."""
    import tempfile
    import time

    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", **({"MFEM_WORKER_TRANSPORT": transport} if transport else {}))
    logs = [tempfile.TemporaryFile() for _ in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "multirank_worker.py"), case, str(world), str(r), str(port)],
                              stdout=logs[r], stderr=subprocess.STDOUT, env=env) for r in range(world)]
    deadline = time.monotonic() + timeout
    try:
        for p in procs:
            try:
                p.wait(timeout=max(deadline - time.monotonic(), 0.1))
            except subprocess.TimeoutExpired:
                break
            if p.returncode != 0:  # a dead rank leaves the others waiting in a collective: do not sit out the timeout
                deadline = min(deadline, time.monotonic() + 20)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
                p.wait()
    outs = []
    for f in logs:
        f.seek(0)
        outs.append(f.read().decode(errors="replace"))
        f.close()
    reports = []
    for r, (p, out) in enumerate(zip(procs, outs)):
        line = [ln for ln in out.splitlines() if ln.startswith("MULTIRANK_REPORT ")]
        rep = json.loads(line[-1][len("MULTIRANK_REPORT "):]) if line else None
        reports.append(rep)
        failed = [] if rep is None else [k for k, v in rep["checks"].items() if not v["ok"]]
        assert p.returncode == 0 and rep is not None and not failed, (
            f"rank {r} of {world} ({case}): rc={p.returncode}, failed checks {failed}\n" +
            (json.dumps({k: rep['checks'][k] for k in failed}, indent=1) if rep else "") + "\n--- output tail ---\n" + out[-3000:])
    return reports


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("case", ["thermal_hex8", "thermal_hex8_small", "elasticity_hex8", "thermal_hex27"])
def test_multirank_solve_equals_single_rank(case, world):
    reports = run_ranks(case, world)
    names = set(reports[0]["checks"])
    for need in ("halo_ghost_planes_bitwise", "jacobi_diag_with_halo_bitwise", "jacobi_colnorm_is_global", "cg_classic_overlap",
                 "cg_classic_blocking", "cg_single_reduction_overlap", "bicgstabl2_diag", "bicgstabl2_colnorm", "idrs8_diag",
                 "idrs8_colnorm", "idrs8_generated_sign_shadows_on_slabs", "callbacks_ran"):
        assert need in names, need
    if case == "thermal_hex8":
        assert "symmetric_sweep_kernel_ran" in names


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("case", ["nitsche_hex8", "nitsche_hex27", "asym_one_slab_hex8"])
def test_multirank_nonsymmetric_solves(case, world):
    """Round 5: the reference's nonsymmetric matrices on slabs.  nitsche_*: Nitsche face on x = 0 -- rank 0's lattice tiles carry a skew remainder
    (spmv_rem.hip), the others none; bicgstabl_GS!(2) / idrs!(8) with Pr_Jacobi! equal the single-rank solve of the global problem (<= 1e-8, same
    shadow vectors).  asym_one_slab: rank 0's tiles REFUSE its values while the other ranks keep theirs -- the rank-local start-over of mfem_solve
    issues no collective (it used to repeat the n_global all-reduce and pair it with the other ranks' next collective)."""
    reports = run_ranks(case, world)
    names = set(reports[0]["checks"])
    for need in ("bicgstabl2_diag_overlap", "idrs8_diag_overlap", "bicgstabl2_diag_blocking", "idrs8_diag_blocking"):
        assert need in names, need


def _n_gpus():
    import torch

    return torch.cuda.device_count()  # (counting devices does not initialise the GPU in this process)


@pytest.mark.skipif(_n_gpus() < 2, reason="the RCCL transport needs one GPU per rank: runs the first time the suite meets a box with >= 2 GPUs")
@pytest.mark.parametrize("case", ["thermal_hex8", "elasticity_hex8", "thermal_hex27"])
def test_two_ranks_over_real_rccl_equal_the_single_rank_solve(case):
    """The library's RCCL transport with MORE THAN ONE rank (ncclSend / ncclRecv of the halo planes on the halo stream beside the interior rows, one
    ncclAllReduce per reduction group): every check of the host-transport runs above -- ghost planes bitwise, Jacobi vectors, every solver against
    the single-rank solve of the global problem -- plus the exposed-communication timers.  One GPU per rank."""
    reports = run_ranks(case, 2, transport="rccl")
    names = set(reports[0]["checks"])
    for need in ("halo_ghost_planes_bitwise", "cg_classic_overlap", "cg_single_reduction_overlap", "bicgstabl2_diag", "idrs8_diag",
                 "rccl_exposed_communication_is_timed", "graph_with_rccl_cg", "graph_with_rccl_idrs8", "graph_with_rccl_cycles_were_captured"):
        assert need in names, need


@pytest.mark.skipif(_n_gpus() < 2, reason="needs 2 GPUs")
@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_two_gpus_over_rccl(scaling):
    """bench.py --gpus 2 --n 64 on two GPUs over RCCL (self-launched child processes), weak and strong scaling: the line carries the per-rank
    exposed communication and the residual check of the timed solve passed (bench.py exits non-zero otherwise)."""
    import subprocess

    root = os.path.dirname(HERE)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--n", "64", "--steps", "2", "--warmup", "1", "--cpu-n", "0",
                        "--scaling", scaling], capture_output=True, text=True, timeout=600, cwd=root,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][-1])
    assert line["comm_exposed"]["exposed_fraction_of_solve"] >= 0  # (the line carries the worst rank; the side file every rank)
    out = json.load(open(os.path.join(root, line["full"])))
    assert out["n_gpus"] == 2 and out["scaling"] == scaling and "RCCL" in out["config"]["parallelism"]
    assert out["config"]["n_dof"] == (129 if scaling == "weak" else 65) * 65 * 65
    assert out["config"]["final_res"] < out["config"]["initial_res"]
    ce = out["comm_exposed"]
    assert len(ce) == 2 and all(c["allreduces_per_step"] > 0 and c["halo_waits_per_step"] > 0 for c in ce)
