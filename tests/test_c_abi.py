"""A plain-C consumer of include/metafem_mi355x.h: tests/c_abi_smoke.c is compiled by gcc (-std=c99, no HIP headers, no Python
mirror of the structs) and linked to libmetafem_mi355x.so.  CPU: it compiles, links and -- without a GPU -- fails loudly with the
library's error message.  GPU (-m gpu): it assembles and solves the 4 x 4 x 4 hex-8 thermal fixture through the three seams and
matches the oracle's K, R0 and T embedded as C arrays (tests/golden/oracle_thermal_hex8_4x4x4.h, written by make_golden.py)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "metafem.jl_amd")
SRC = os.path.join(ROOT, "tests", "c_abi_smoke.c")


def _build(tmp_path):
    exe = str(tmp_path / "c_abi_smoke")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-O1", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "tests"), SRC, "-o", exe, "-L" + LIBDIR, "-l:libmetafem_mi355x.so",
                           "-L/opt/rocm/lib", "-lamdhip64", "-lm", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_c_header_fixture_is_the_npz_fixture():
    """The C arrays are the committed .npz fixture, value for value (hexadecimal floating constants: exact)."""
    import re

    d = np.load(os.path.join(ROOT, "tests", "golden", "oracle_thermal_hex8_4x4x4.npz"))
    text = open(os.path.join(ROOT, "tests", "golden", "oracle_thermal_hex8_4x4x4.h")).read()
    for name, key in (("gold_K", "K"), ("gold_R0", "R0"), ("gold_T", "T")):
        body = re.search(name + r"\[\d+\] = \{(.*?)\};", text, flags=re.S).group(1)
        vals = np.array([float.fromhex(v.strip()) for v in body.split(",")])
        assert np.array_equal(vals, d[key]), name
    body = re.search(r"gold_colidx\[\d+\] = \{(.*?)\};", text, flags=re.S).group(1)
    assert np.array_equal(np.array([int(v) for v in body.split(",")]), d["colidx"])


def test_c_consumer_compiles_links_and_fails_loudly_without_a_gpu(tmp_path):
    import torch

    exe = _build(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("GPU present: the run itself is test_c_consumer_runs_the_three_seams")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "mfem_context_create" in r.stderr and "C_ABI_SMOKE OK" not in r.stdout


@pytest.mark.gpu
def test_c_consumer_runs_the_three_seams(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "C_ABI_SMOKE OK" in r.stdout, r.stdout[-3000:] + "\n" + r.stderr[-3000:]
