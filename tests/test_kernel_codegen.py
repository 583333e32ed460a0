"""CPU only: guard rails on the compiled query-stationary search kernel (csrc/search_qs.hip).

Its tile loop runs with every register spoken for (Q fragments: 256 AGPRs + 128 VGPRs per lane) and a deep LDS-DMA ring in
flight.  Two things the compiler can do there cost far more than they look:
  * a scratch (spill) reload inside the loop: hipcc waits `vmcnt(0)` for it, which drains the whole DMA ring -- the
    selection path once took 5 us per entry that way;
  * shuffling the stationary Q fragments between the accumulator file and arch VGPRs (`v_accvgpr_*`) every k-step.
Both have happened after innocent-looking source edits, with correct results and a 10-30 % slower kernel, so the
assembly is checked."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "multimodal_rag_amd", "csrc", "search_qs.hip")


def _kernels(asm: str):
    cur, name = None, None
    for line in asm.splitlines():
        m = re.match(r"^(_ZN10mmrag_impl21cosine_topk_qs_kernel\w+):", line)
        if m:
            name, cur = m.group(1), []
        if cur is not None:
            cur.append(line)
            if "s_endpgm" in line:
                yield name, cur
                cur = None


def test_tile_loop_has_no_spills_and_no_accvgpr_traffic(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = str(tmp_path / "search_qs.s")
    r = subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-I",
                        os.path.join(ROOT, "include"), SRC, "-o", out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    seen = 0
    for name, lines in _kernels(open(out).read()):
        seen += 1
        mfma = [i for i, l in enumerate(lines) if "v_mfma_f32" in l]
        assert mfma, name
        labels = {m.group(1): i for i, l in enumerate(lines) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
        # the tile loop: the last backward branch after the last MFMA whose target sits before the first MFMA
        back = [(i, labels[m.group(1)]) for i, l in enumerate(lines)
                if i > mfma[-1] and (m := re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)) and m.group(1) in labels
                and labels[m.group(1)] < mfma[0]]
        assert back, f"{name}: tile loop not found"
        end, start = back[-1]
        body = lines[start:end + 1]
        in_k_steps = lines[mfma[0]:mfma[-1] + 1]
        assert sum("v_mfma_f32" in l for l in body) % 192 in (0, 96, 128), name      # one copy of the unrolled tile
        for l in body:
            assert "scratch_" not in l, f"{name}: spill access inside the tile loop: {l.strip()}"
        for l in in_k_steps:
            assert "v_accvgpr" not in l, f"{name}: Q fragments move between register files in the k-steps: {l.strip()}"
            assert not re.search(r"s_waitcnt.*vmcnt\(0\)", l) or "ASM" in l, f"{name}: vmcnt(0) among the k-steps"
    assert seen >= 12   # 2 dtypes x 3 row lengths x 2 list depths (x 2 cache policies)


def _dest_regs(line: str):
    """VGPRs an instruction line writes (first operand), as a set of register numbers"""
    m = re.match(r"\s+[a-z_0-9]+\s+v(\d+)\b", line)
    if m:
        return {int(m.group(1))}
    m = re.match(r"\s+[a-z_0-9]+\s+v\[(\d+):(\d+)\]", line)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def test_walk_kernel_tile_loop(tmp_path):
    """csrc/search_qsw.hip (the single-launch walk): the same two guard rails among its k-steps, and the ticket
    register: the returning atomic of the dynamic tile hand-out is inline asm (hipcc would wait vmcnt(0) for a
    builtin's result at once), its value lands in the register some time during the tile and is read only by the asm
    store at the tile's end -- nothing may write or move that register in between."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = str(tmp_path / "search_qsw.s")
    r = subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-I",
                        os.path.join(ROOT, "include"), os.path.join(ROOT, "multimodal_rag_amd", "csrc", "search_qsw.hip"),
                        "-o", out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    asm = open(out).read()
    kernels = re.findall(r"^(_ZN10mmrag_impl23cosine_topk_walk_kernel\w+):[^\n]*\n(.*?)s_endpgm", asm, re.S | re.M)
    assert len(kernels) >= 12          # 2 dtypes x 3 row lengths x 2 cache policies
    for name, body in kernels:
        lines = body.splitlines()
        mfma = [i for i, l in enumerate(lines) if "v_mfma_f32" in l]
        assert mfma and len(mfma) % 64 == 0, name      # 32 MFMAs per 128-byte K-slab of a tile
        for l in lines[mfma[0]:mfma[-1] + 1]:
            assert "scratch_" not in l, f"{name}: spill access among the k-steps: {l.strip()}"
            assert "v_accvgpr" not in l, f"{name}: Q fragments move between register files: {l.strip()}"
            assert not re.search(r"s_waitcnt.*vmcnt\(0\)", l), f"{name}: vmcnt(0) among the k-steps"
        atom = [i for i, l in enumerate(lines) if "global_atomic_add" in l]
        assert len(atom) == 1, name
        reg = int(re.search(r"global_atomic_add\s+v(\d+),", lines[atom[0]]).group(1))
        store = [i for i in range(atom[0], len(lines))
                 if re.search(r"ds_write_b32\s+v\d+,\s+v%d\b" % reg, lines[i]) and "exec, 1" in lines[i - 1]]
        assert store, f"{name}: the ticket store does not read v{reg}"
        for l in lines[atom[0] + 1:store[0]]:
            assert reg not in _dest_regs(l), f"{name}: v{reg} (ticket in flight) is written by: {l.strip()}"
            assert "scratch_" not in l, f"{name}: spill access inside the tile loop: {l.strip()}"


def test_persistent_linear_kernel_has_no_scratch(tmp_path):
    """csrc/encoder.hip, linear_persistent_kernel: its K loop carries 128 accumulator registers and two fragment sets
    with an LDS-DMA ring in flight; a spill reload inside it waits vmcnt(0) and stalls the ring (seen with 16 waves of
    128 registers: +15 %).  The 8-wave form must compile without any scratch."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = str(tmp_path / "encoder.s")
    r = subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-I",
                        os.path.join(ROOT, "include"), os.path.join(ROOT, "multimodal_rag_amd", "csrc", "encoder.hip"),
                        "-o", out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    asm = open(out).read()
    body = re.search(r"^(_ZN10mmrag_impl24linear_persistent_kernel\w+):[^\n]*\n(.*?)s_endpgm", asm, re.S | re.M)
    assert body, "linear_persistent_kernel not found"
    assert "scratch_" not in body.group(2)
    meta = re.search(r"\.name:\s+_ZN10mmrag_impl24linear_persistent_kernel.*?\.vgpr_spill_count:\s+(\d+)", asm, re.S)
    assert meta and int(meta.group(1)) == 0
