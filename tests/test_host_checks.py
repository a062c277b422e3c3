"""CPU checks of host/device headers of the product, compiled with g++ (no GPU): the sum-factorised hex-8 element routines against the
stored-table form (reference semantics: 4_Update_Integrator.jl:2-33,90-154; 06_FEM_Kernel.jl:28-45,65-79) and the LDS mirror-table /
edge-block bookkeeping of the wave-private symmetric sweep, replayed on small lattices; the step / slot tables of the symmetric lattice-tile layouts
(every stencil pair listed exactly once: a product through the tables against the entry-by-entry product)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", ["host_check_hex8", "host_check_symp", "host_check_lat"])
def test_host_check(name, tmp_path):
    exe = str(tmp_path / name)
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "metafem.jl_amd", "csrc"),
                    os.path.join(ROOT, "tools", name + ".cpp"), "-o", exe], check=True)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:]
    assert out.stdout.strip().endswith("OK")
