"""bench.py end to end on one GPU, as child processes (started before they touch the GPU; never re-exec'ed): the single-GPU line, the
N > 1 launch path with the ranks sharing cuda:0 through the host-callback communicator (functional, not a measurement), the RCCL path
with world = 1 (communicator, ring self-test, split SpMV schedule), and the c3 / c4 legs."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--steps", "1", "--warmup", "0", "--secondary-n", "0", "--cpu-n", "0", "--hex27-n", "0"]


def run_bench(args, env=None, timeout=900):
    """-> the FULL result object (the side file the line names); the stdout line itself is checked here for every invocation: ONE line, strict JSON,
    under 8 KB, contract keys + roofline + cpu_baseline (what the driver parses -- round 5's 32 KB line was not)."""
    import tempfile

    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **(env or {}))
    with tempfile.TemporaryDirectory() as tmp:
        side = os.path.join(tmp, "full.json")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--full-out", side] + args, capture_output=True, text=True, timeout=timeout,
                           env=e, cwd=ROOT)
        lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
        assert r.returncode == 0, f"rc={r.returncode}\n--- stdout ---\n{r.stdout[-2000:]}\n--- stderr ---\n{r.stderr[-4000:]}"
        assert len(lines) == 1, f"stdout must carry ONE JSON line, got {len(lines)}:\n{r.stdout[-2000:]}"
        assert len(lines[0].encode()) < 8192, len(lines[0])

        def no_constants(c):
            raise ValueError(c)
        line = json.loads(lines[0], parse_constant=no_constants)
        full = json.load(open(side))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["full"] == side and abs(line["value"] / full["value"] - 1) < 1e-4
    assert 0 < line["roofline"]["frac"] < 1 and abs(line["roofline"]["frac"] - full["roofline"]["frac"]) < 1e-4 and len(line["roofline"]["kernel"]) <= 100
    for k in ("workload", "n_dof", "nnz", "solve_ms_per_step", "assembly_ms_per_step", "initial_res", "final_res"):
        assert k in line["config"], k
    assert "errors" not in full, full["errors"]
    full["_line"] = line
    return full


def _common(out, n_gpus, steps=1, scaling="weak"):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in out, k
    assert out["n_gpus"] == n_gpus and out["steps"] == steps and out["value"] > 0 and out["scaling"] == scaling
    assert out["dtype"] == "f64" and out["vs_baseline"] is None and "workload" in out["config"]
    rf = out["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert 0 < rf["frac"] < 1.0
    # bench.py checks what it times: the residuals of the last timed solve are in the line (and the run fails when the solve did not reduce them)
    c = out["config"]
    assert c["initial_res"] > 0 and c["final_res"] >= 0 and c["final_res"] == c["final_res"] and "workspace_placement_trial" in c


def test_single_gpu_line_has_every_object():
    out = run_bench(["--n", "40", "--steps", "2", "--warmup", "1", "--secondary-n", "24", "--secondary-steps", "1", "--hex27-n", "8", "--cpu-n", "12",
                     "--cpu-repeats", "1", "--iters", "30"])
    _common(out, 1, steps=2)
    assert out["config"]["n_dof"] == 41 ** 3 and out["config"]["krylov_steps_per_step"] == 30
    assert out["config"]["final_res"] < out["config"]["initial_res"] and out["config"]["first_step_s"] > 0
    assert out["config"]["workspace_placement_trial"].startswith("on")  # bench.py opts in explicitly (the library's default is off)
    ck = out["roofline"]["csr_kernel"]
    # (41-point lattice lines: no 64-row tile sits inside one line, so nothing is elided here; + the tile table)
    assert ck["column_entries_read"] <= ck["nnz"] and ck["algorithmic_bytes_per_launch"] <= 1.001 * ck["csr_equivalent"]["bytes_per_launch"]
    assert "frac_actual" in ck and "traffic_over_algorithmic" in ck
    assert "csr_kernel_hex27" in out["roofline"] and "error" not in out["roofline"]["csr_kernel_hex27"]
    sec = out["secondary_24"]
    assert sec["n_dof"] == 25 ** 3 and sec["value"] > 0 and 0 < sec["roofline"]["frac"] < 1 and sec["final_res"] < sec["initial_res"]
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert out["vs_cpu_baseline"]["main_workload"] > 0
    # the driver's line: cpu_baseline as numbers + a sample of at most 120 characters, one small object per leg
    line = out["_line"]
    assert line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["cores"] == cb["cores"] and len(line["cpu_baseline"]["sample"]) <= 120
    assert set(line["secondary_24"]) >= {"value", "ms_per_step", "spmv_ms", "frac", "frac_actual", "csr_frac"}
    assert line["roofline"]["csr_kernel"]["ms"] > 0
    nl = out["newton_like"]
    assert nl["iterations"] == 20 and nl["ms_per_step"] > 0 and nl["solve_ms"] > 0 and line["newton_like"]["per_solve_ms"] == pytest.approx(nl["per_solve_ms"], rel=1e-3)


def test_two_ranks_on_one_gpu_through_the_host_transport():
    out = run_bench(["--gpus", "2", "--n", "48"] + SMALL, env={"MFEM_BENCH_HOST_COMM": "1"})
    _common(out, 2)
    assert out["config"]["n_dof"] == 97 * 49 * 49 and "slab decomposition x2" in out["config"]["parallelism"]
    assert out["cpu_baseline"] is None
    # round 5: the weak line also carries the STRONG reading (the config's own mesh cut into N slabs), so one scaling run answers both
    ss = out["strong_scaling"]
    assert ss["scaling"] == "strong" and ss["n_dof"] == 49 ** 3 and ss["value"] > 0 and ss["final_res"] < ss["initial_res"]
    assert [c["rank"] for c in ss["comm_exposed"]] == [0, 1]
    # the communication each rank's solver stream was exposed to, per rank (host transport: the whole staged exchange is exposed)
    ce = out["comm_exposed"]
    assert [c["rank"] for c in ce] == [0, 1]
    for c in ce:
        assert c["halo_waits_per_step"] >= 200 and c["allreduces_per_step"] >= 200 and c["halo_wait_ms_per_step"] > 0 and c["allreduce_ms_per_step"] > 0
        assert 0 < c["exposed_fraction_of_solve"] < 1.5


def test_strong_scaling_cuts_the_one_gpu_mesh_into_slabs():
    """--scaling strong: THE n^3 mesh is cut into N slabs (north_star: 'the mesh is domain-decomposed across the 8 GPUs'), the line says
    "strong" and the DOF count is the one-GPU mesh's; same for the vector and the hex-27 configs (slabs of an order-2 lattice start on element
    boundaries)."""
    out = run_bench(["--gpus", "2", "--n", "48", "--scaling", "strong"] + SMALL, env={"MFEM_BENCH_HOST_COMM": "1"})
    _common(out, 2, scaling="strong")
    assert out["config"]["n_dof"] == 49 ** 3 and "cut into 2 slabs" in out["config"]["workload"]
    assert out["config"]["final_res"] < out["config"]["initial_res"]
    for config, n, iters, ndof in (("c3", 16, 40, 3 * 17 ** 3), ("c4", 8, 20, 17 ** 3)):
        out = run_bench(["--config", config, "--gpus", "2", "--n", str(n), "--iters", str(iters), "--scaling", "strong"] + SMALL,
                        env={"MFEM_BENCH_HOST_COMM": "1"})
        _common(out, 2, scaling="strong")
        assert out["config"]["n_dof"] == ndof


def test_default_line_carries_all_four_configs():
    """The default invocation (c2, N = 1) also runs configs[2] and configs[3] and the FP64-MFMA roofline of the hex-27 Ke kernels, so the driver's
    one line carries all four configs; here at reduced sizes (--secondary-config-n)."""
    out = run_bench(["--n", "32", "--steps", "1", "--warmup", "1", "--secondary-n", "0", "--hex27-n", "0", "--cpu-n", "0", "--iters", "40",
                     "--secondary-config-n", "12", "--secondary-steps", "2"])
    _common(out, 1)
    for key, ndof, metric in (("secondary_c3", 3 * 13 ** 3, "elasticity"), ("secondary_c4", 25 ** 3, "hex-27")):
        t = out[key]
        assert t["n_dof"] == ndof and t["value"] > 0 and t["steps"] == 2 and metric in t["metric"]
        assert 0 < t["roofline"]["frac"] < 1 and t["csr_kernel"]["frac"] > 0
        assert t["final_res"] == t["final_res"] and t["initial_res"] > 0
    # round 5: the c3 leg also says what solving that system TO A TOLERANCE costs with the reference's solvers and with cg! ...
    ttt = out["secondary_c3"]["time_to_tol"]
    for name in ("bicgstabl_GS!(2)", "idrs!(8)", "cg!"):
        assert ttt[name]["iterations"] > 0 and ttt[name]["ms"] > 0 and ttt[name]["converged"] is True and ttt[name]["final_res_over_initial"] <= 1e-8
    # ... and the line carries the reference's own solver / boundary-condition path: idrs!(8) on configs[1], the Nitsche-Dirichlet NONSYMMETRIC K on
    # configs[1] and configs[3] under bicgstabl_GS!(2) (reduced sizes here)
    for key, ndof in (("ref_idrs8_256", 13 ** 3), ("nitsche_c2_256", 13 ** 3), ("nitsche_c4_128", 25 ** 3)):
        t = out[key]
        assert t["n_dof"] == ndof and t["value"] > 0 and t["steps"] == 2 and 0 < t["roofline"]["frac"] < 1
        assert t["initial_res"] > 0 and t["final_res"] < 2.0 * t["initial_res"]
        assert abs(t["final_res_recomputed"] - t["final_res"]) <= 1e-5 * t["final_res"] + 1e-9 * t["initial_res"]
    assert out["nitsche_c2_256"]["csr_kernel"]["frac"] > 0
    ke = out["roofline_hex27_ke"]
    assert ke["bound"] == "mfma" and ke["peak"] == 78.6 and ke["useful_flop_per_assembly"] == 118098.0 * 12 ** 3
    assert ke["achieved"] > 0 and 0 < ke["frac"] < 1 and abs(ke["frac"] - ke["achieved"] / ke["peak"]) < 1e-12
    # round 5: the same mesh with every element distorted, assembled by the row-owner kernel of general elements (the line fails unless it ran and agrees
    # with the two-pass MFMA path to 2e-13)
    gm = ke["general_mesh_assembly"]
    assert gm["avg_assembly_ms"] > 0 and gm["two_pass_mfma_path_same_mesh_ms"] > 0 and gm["max_rel_difference_to_the_mfma_path"] <= 2e-13


@pytest.mark.parametrize("config,n,iters,ndof", [("c3", 16, 40, 3 * 33 * 17 * 17), ("c4", 8, 20, 33 * 17 * 17)])
def test_c3_c4_legs_with_two_ranks(config, n, iters, ndof):
    out = run_bench(["--config", config, "--gpus", "2", "--n", str(n), "--iters", str(iters)] + SMALL, env={"MFEM_BENCH_HOST_COMM": "1"})
    _common(out, 2)
    assert out["config"]["n_dof"] == ndof
    assert out["config"]["krylov_steps_per_step"] >= iters * 0.9


@pytest.mark.parametrize("config,n,ndof", [("c3", 24, 3 * 25 ** 3), ("c4", 12, 25 ** 3), ("nitsche_c2", 24, 25 ** 3), ("ref_idrs8", 24, 25 ** 3)])
def test_c3_c4_legs_single_gpu(config, n, ndof):
    out = run_bench(["--config", config, "--n", str(n), "--iters", "40"] + SMALL)
    _common(out, 1)
    assert out["config"]["n_dof"] == ndof and "csr_kernel" in out["roofline"]


def test_traffic_is_collected_in_the_run_at_a_config_size():
    """At a config's own size bench.py observes the HBM counters itself: two rocprofv3 --pmc child passes (FETCH_SIZE, WRITE_SIZE) on the
    workload's leg; `traffic_source` says so and the calibration kernel's known bytes are met.  (c3 at 128^3: the smallest such leg.)"""
    out = run_bench(["--config", "c3", "--steps", "1", "--warmup", "0", "--iters", "40"], timeout=1200)
    _common(out, 1)
    for obj in (out["roofline"], out["roofline"]["csr_kernel"]):
        assert obj["traffic"] and obj["traffic_source"].startswith("collected in this run"), obj.get("traffic_source")
        assert 0.9 < obj["traffic_over_algorithmic"] < 1.4 and 0 < obj["frac_actual"] < 1
        assert "read 1.000" in obj["traffic_source"] and "written 1.000" in obj["traffic_source"]


def test_falls_back_to_the_host_transport_when_the_rccl_self_test_fails():
    """If the library's RCCL transport is not usable (here: simulated), every rank agrees on the host-callback transport and the run still
    prints its line, saying which transport carried it."""
    out = run_bench(["--gpus", "1", "--n", "40"] + SMALL, env={"MFEM_BENCH_FORCE_COMM": "1", "MFEM_BENCH_SIMULATE_RCCL_FAILURE": "1"})
    _common(out, 1)
    assert out["config"]["n_dof"] == 41 ** 3


def test_self_launch_ends_the_other_ranks_when_one_dies():
    """`python bench.py --gpus 2` on a box with ONE GPU and without the host-transport switch: rank 1 has no device and exits with an
    error; the launcher must end rank 0 (which would wait in the rendezvous for ever) and report the failure instead of hanging."""
    import time

    import torch

    if torch.cuda.device_count() != 1:
        pytest.skip("needs a box with exactly one GPU")
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    e.pop("MFEM_BENCH_HOST_COMM", None)
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--n", "24"] + SMALL, capture_output=True, text=True,
                       timeout=300, env=e, cwd=ROOT)
    assert r.returncode != 0 and "ranks failed" in r.stderr, r.stderr[-2000:]
    assert time.time() - t0 < 200
