"""CPU only: the host-side C++ of libmmrag.so (multi-threaded tokenizers, host merge) under AddressSanitizer +
UndefinedBehaviorSanitizer and under ThreadSanitizer.  Never run on the GPU box."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "multimodal_rag_amd", "csrc")
DRIVER = os.path.join(ROOT, "tests", "native", "sanitize_host.cpp")


def _hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.parametrize("name,flags", [
    ("asan_ubsan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]),
    ("tsan", ["-fsanitize=thread"]),
])
def test_host_cpp_under_sanitizers(tmp_path, name, flags):
    hipcc = _hipcc()
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    exe = str(tmp_path / f"sanitize_host_{name}")
    # host-only compile of the two translation units (they include the HIP headers through mmrag_internal.h)
    cmd = [hipcc, "-x", "hip", "--cuda-host-only", "-O1", "-g", "-std=c++17", "-fno-omit-frame-pointer", *flags,
           "-I", os.path.join(ROOT, "include"), os.path.join(CSRC, "tokenizer.cpp"), os.path.join(CSRC, "clip_bpe.cpp"),
           os.path.join(CSRC, "host_merge.cpp"),
           DRIVER, "-o", exe, "-lpthread"]
    build = subprocess.run(cmd, capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1",
               TSAN_OPTIONS="halt_on_error=1")
    run = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=600)
    assert run.returncode == 0 and "sanitize_host: ok" in run.stdout, (run.stdout[-2000:], run.stderr[-4000:])
    assert "ERROR: AddressSanitizer" not in run.stderr and "WARNING: ThreadSanitizer" not in run.stderr
    assert "runtime error" not in run.stderr, run.stderr[-4000:]
