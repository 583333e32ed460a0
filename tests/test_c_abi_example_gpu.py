"""GPU: the torch-free C-ABI example (examples/c_abi_search.cpp) builds against include/mmrag.h + libmmrag.so and
finds every planted row -- the boundary is usable from plain HIP/C++ with no Python or torch in the process."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_example_builds_and_runs(tmp_path):
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    lib = os.path.join(ROOT, "multimodal_rag_amd", "lib")
    assert os.path.exists(os.path.join(lib, "libmmrag.so")), "build the library first (__graft_entry__.build())"
    exe = str(tmp_path / "c_abi_search")
    subprocess.run([hipcc, "-O2", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", "c_abi_search.cpp"), "-L", lib, "-lmmrag", f"-Wl,-rpath,{lib}",
                    "-o", exe], check=True, timeout=300)
    out = subprocess.run([exe, "60000", "48"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "planted rows found first: 48 / 48" in out.stdout
