"""The ONE stdout line of bench.py (bench_report.compact_line) and the --dry mode, on the CPU.

Round 5's driver record was UNPARSED: the line had grown to 32 KB.  The line is now built from the full result object by a pure function with a hard size
bound; here it runs on canned objects -- among them round 5's committed 32 KB line -- and `bench.py --gpus 8 --dry 1` exercises everything an 8-rank run
does on the host (rank spawn, rendezvous, slab planes, halo sizes, device bytes per rank, max-over-ranks, the line) without a device."""
import copy
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench_report as br  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
            "roofline", "cpu_baseline")


def strict_loads(line):
    def no_constants(c):
        raise ValueError(f"non-JSON constant {c}")
    return json.loads(line, parse_constant=no_constants)


def r05_full():
    return json.load(open(os.path.join(ROOT, "profiles", "r05_bench_n1.json")))


def check_line(line):
    assert "\n" not in line and len(line.encode()) < br.MAX_LINE_BYTES
    d = strict_loads(line)
    for k in CONTRACT:
        assert k in d, k
    return d


def test_round5_object_fits_and_keeps_the_numbers():
    full = r05_full()
    assert len(json.dumps(full)) > 30000  # (the line the driver could not parse)
    d = check_line(br.compact_line(full, "gpurun_out/bench_full.json"))
    assert abs(d["value"] / full["value"] - 1) < 1e-4 and d["full"] == "gpurun_out/bench_full.json"
    rf = d["roofline"]
    assert abs(rf["frac"] - full["roofline"]["frac"]) < 1e-4 and rf["bound"] == "hbm" and rf["peak"] == 8000.0 and rf["unit"] == "GB/s"
    assert abs(rf["achieved"] / rf["peak"] - rf["frac"]) < 1e-4 and rf["traffic"] > 0 and len(rf["kernel"]) <= 100
    assert rf["csr_kernel"]["ms"] > 0 and 0 < rf["csr_kernel"]["frac"] < 1
    cb = d["cpu_baseline"]
    assert cb["value"] == pytest.approx(full["cpu_baseline"]["value"], rel=1e-4) and cb["cores"] == 16 and cb["kind"] == "port" and len(cb["sample"]) <= 120
    for leg in ("secondary_256", "secondary_c3", "secondary_c4", "ref_idrs8_256", "nitsche_c2_256", "nitsche_c4_128"):
        assert set(d[leg]) >= {"value", "ms_per_step", "spmv_ms", "frac", "frac_actual", "csr_frac"} and d[leg]["value"] > 0
        assert len(json.dumps(d[leg])) < 300
    assert d["roofline_hex27_ke"]["bound"] == "mfma" and 0 < d["roofline_hex27_ke"]["frac"] < 1
    c = d["config"]
    for k in ("workload", "n_dof", "nnz", "solve_ms_per_step", "assembly_ms_per_step", "initial_res", "final_res"):
        assert k in c, k


def test_no_object_can_unparse_the_line_again():
    """Whatever a future leg puts into the full object -- prose, many legs, NaN -- the line stays under 8 KB, strict JSON, with the contract keys."""
    full = r05_full()
    full["roofline"]["kernel"] = "x" * 20000
    full["config"]["workload"] = "w" * 5000
    full["cpu_baseline"]["sample"] = "s" * 9000
    full["metric"] = "m" * 300
    for i in range(60):
        full[f"secondary_extra_{i}"] = copy.deepcopy(full["secondary_c3"])
        full[f"secondary_extra_{i}"]["roofline"]["kernel_key"] = "k" * 500
    full["secondary_c3"]["value"] = float("nan")
    full["roofline"]["frac_actual"] = float("inf")
    d = check_line(br.compact_line(full, "f.json"))
    assert d["secondary_c3"]["value"] is None and d["roofline"]["frac_actual"] is None
    assert d["dropped"] and all(k.startswith("secondary_extra_") for k in d["dropped"])  # the newest legs go first; the named ones stay
    assert "secondary_256" in d and "roofline_hex27_ke" in d


def test_a_failed_leg_is_an_error_object():
    full = r05_full()
    full["secondary_c3"] = {"error": "ResidualCheckFailed: c3 128^3: the last timed solve did not reduce the residual " + "z" * 1000}
    full["cpu_baseline"] = {"error": "OSError: liboracle.so"}
    full["errors"] = ["secondary_c3: ...", "cpu_baseline: ..."]
    d = check_line(br.compact_line(full))
    assert set(d["secondary_c3"]) == {"error"} and len(d["secondary_c3"]["error"]) <= 200
    assert set(d["cpu_baseline"]) == {"error"} and d["value"] > 0 and len(d["errors"]) == 2


def test_multi_rank_objects_are_summarised():
    full = r05_full()
    ce = [{"rank": r, "halo_wait_ms_per_step": 0.1 * r, "halo_waits_per_step": 200.0, "allreduce_ms_per_step": 1.0, "allreduces_per_step": 200.0,
           "solve_ms_per_step": 100.0, "exposed_fraction_of_solve": 0.01 + 0.001 * r} for r in range(8)]
    full["comm_exposed"] = ce
    full["strong_scaling"] = {"scaling": "strong", "value": 1e11, "n_dof": 135005697, "ms_per_step": 200.0, "solve_ms_per_step": 190.0, "comm_exposed": ce,
                              "workload": "y" * 500}
    d = check_line(br.compact_line(full))
    assert d["comm_exposed"]["rank"] == 7 and d["strong_scaling"]["value"] == 1e11 and d["strong_scaling"]["comm_exposed"]["rank"] == 7


def run_dry(extra, world):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--dry", "1", "--full-out", "-"] + extra, capture_output=True,
                       text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    return check_line(lines[0])


def test_dry_run_of_the_drivers_eight_gpu_command():
    """`bench.py --gpus 8` as the driver starts it, minus the device: 8 child ranks spawned before any GPU call, gloo rendezvous, the weak plan (512^3 per rank)
    and the strong plan (512^3 in all), each slab inside 288 GB."""
    d = run_dry([], 8)
    assert d["dry"] is True and d["n_gpus"] == 8 and d["scaling"] == "weak" and d["value"] is None
    assert d["config"]["n_dof"] == (8 * 512 + 1) * 513 * 513
    w, s = d["dry_plan"]["weak"], d["dry_plan"]["strong"]
    assert w["ranks"] == 8 and s["ranks"] == 8 and w["fits_288GB"] and s["fits_288GB"]
    assert w["halo_bytes"] == 513 * 513 * 8 and s["halo_bytes"] == 513 * 513 * 8
    assert w["max_rows"] == 513 * 513 * 513 and w["min_rows"] == 512 * 513 * 513  # (4097 planes: the first rank takes the extra one)
    assert s["max_rows"] == 65 * 513 * 513 and s["min_rows"] == 64 * 513 * 513
    assert 50e9 < w["max_est_device_bytes"] < 0.9 * 288e9


@pytest.mark.parametrize("config,scaling,world", [("c3", "strong", 4), ("c4", "strong", 3), ("c4", "weak", 2)])
def test_dry_run_other_configs(config, scaling, world):
    d = run_dry(["--config", config, "--scaling", scaling], world)
    assert d["dry"] and d["scaling"] == scaling and d["dry_plan"][scaling]["ranks"] == world and d["dry_plan"][scaling]["fits_288GB"]
    if config == "c4":  # two ghost planes per neighbour (order 2)
        assert d["dry_plan"][scaling]["halo_bytes"] == 2 * 257 * 257 * 8


def test_dry_plan_nnz_is_the_lattice_stencil():
    """The closed form of bench_dry equals the known totals of the full meshes (256^3 hex-8 nnz = 454 756 609; 512^3: 3 630 961 153)."""
    sys.path.insert(0, ROOT)
    import bench_dry

    for n, nnz in ((256, 454756609), (512, 3630961153)):
        cfg = dict(order=1, fields=1, solver="cg")
        assert sum(bench_dry.slab_plan(cfg, n, 4, r, True)["nnz"] for r in range(4)) == nnz
    hex27 = sum(bench_dry.slab_plan(dict(order=2, fields=1, solver="cg"), 8, 2, r, True)["nnz"] for r in range(2))
    # hex-27 8^3: per direction 9 corner points (2 x 3 + 7 x 5) + 8 mid points x 3 = 65 couplings
    assert hex27 == 65 ** 3
