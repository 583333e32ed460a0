"""Developer probe: shader cycles per v_mfma_f32_16x16x32_f16 and the clock held, for loops that add the pieces of the
search walk's k-step one at a time (csrc/microbench.hip: mfma_probe_kernel).  Each mode is held ~1 s before it is read."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_rag_amd import _native as N
L = N.lib()
L.mmrag_internal_mfma_probe.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
seed = (torch.randn(256 * 8, device="cuda") * 0.5).half()
ncu = torch.cuda.get_device_properties(0).multi_processor_count
out = torch.empty(ncu * 256, device="cuda"); stamps = torch.zeros(ncu * 2, dtype=torch.int64, device="cuda")
iters = 20000
names = ["A,B in VGPRs", "B in AGPRs", "B rotating over 64 AGPRs", "+ v_xad + 2 ds_read_b128 / 8 MFMAs", "+ s_waitcnt lgkmcnt(0)",
         "+ one LDS-DMA piece / 16 MFMAs (per-lane offsets)", "+ one LDS-DMA piece / 16 MFMAs (no per-lane offset)",
         "+ two LDS-DMA pieces back to back / 32 MFMAs", "mode 4 with its fillers spread over the MFMA gaps", "+ one global_load_lds piece / 16 MFMAs"]
st = torch.cuda.current_stream().cuda_stream
for rnd in range(2):
    for mode in range(10):
        for _ in range(int(os.environ.get("HOLD", "200"))):
            rc = L.mmrag_internal_mfma_probe(seed.data_ptr(), out.data_ptr(), iters, mode, stamps.data_ptr(), st); assert rc == 0
        torch.cuda.synchronize()
        s = stamps.cpu().view(ncu, 2).double()
        cyc = (s[:, 0] / (iters * 32)).median().item(); mhz = (s[:, 0] / s[:, 1] * 100).median().item()
        us = (s[:, 1] / 100).median().item()
        tf = ncu * 4 * iters * 32 * 2 * 16 * 16 * 32 / (us * 1e-6) / 1e12
        print(f"mode {mode} ({names[mode]}): {cyc:.2f} cycles/MFMA at {mhz:.0f} MHz -> {tf:.0f} TFLOP/s", flush=True)
