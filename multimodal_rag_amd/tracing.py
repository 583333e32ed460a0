"""Per-stage timers and roctx ranges of the embed / retrieve path.

The reference only logs `time.time()` deltas (embedder.py:452, 492-498; api.py:251, 302, 398).  Here every stage of a
request -- tokenize, encode (device forward + copy-out), search (kernel enqueue), collect (device -> host + result
building), and in the sharded service broadcast / exchange / merge -- runs inside `stage(name)`:

  * always: wall-clock seconds, call count and maximum per stage (`snapshot()`, shown by
    `EmbeddingManager.get_collection_stats()["stages"]`; `reset()` clears them);
  * with `MMRAG_ROCTX=1` in the environment: a roctx range of the same name (librocprofiler-sdk-roctx), so that
    `rocprofv3 --marker-trace --kernel-trace -- python ...` shows which kernels belong to which stage.

Host-side bookkeeping only: no device synchronisation is added, so a stage that merely enqueues kernels ("search")
reports its enqueue time and the wait shows up in the stage that reads the result ("collect").
"""
from __future__ import annotations

import contextlib
import ctypes
import os
import threading
import time
from typing import Dict

_lock = threading.Lock()
timeline = None   # developer hook: a list collects (thread, stage, start, end) of every stage (tools/served_callers.py)
_stages: Dict[str, list] = {}      # name -> [calls, total seconds, max seconds]
_roctx = None
_roctx_tried = False


def _load_roctx():
    global _roctx, _roctx_tried
    if _roctx_tried:
        return _roctx
    _roctx_tried = True
    if os.environ.get("MMRAG_ROCTX", "0").lower() not in ("1", "true", "yes"):
        return None
    # rocprofv3 records the rocprofiler-sdk flavour of ROCTx; the roctracer one (libroctx64) serves older profilers
    for name in ("librocprofiler-sdk-roctx.so", "/opt/rocm/lib/librocprofiler-sdk-roctx.so", "libroctx64.so",
                 "/opt/rocm/lib/libroctx64.so"):
        try:
            lib = ctypes.CDLL(name)
            lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
            lib.roctxRangePushA.restype = ctypes.c_int
            lib.roctxRangePop.restype = ctypes.c_int
            _roctx = lib
            break
        except OSError:
            continue
    return _roctx


def roctx_enabled() -> bool:
    return _load_roctx() is not None


@contextlib.contextmanager
def stage(name: str):
    lib = _load_roctx()
    if lib is not None:
        lib.roctxRangePushA(name.encode())
    t0 = time.perf_counter()
    try:
        yield
    finally:
        dt = time.perf_counter() - t0
        if lib is not None:
            lib.roctxRangePop()
        if timeline is not None:
            timeline.append((threading.get_ident(), name, t0, t0 + dt))
        with _lock:
            rec = _stages.get(name)
            if rec is None:
                _stages[name] = [1, dt, dt]
            else:
                rec[0] += 1
                rec[1] += dt
                if dt > rec[2]:
                    rec[2] = dt


def snapshot() -> Dict[str, Dict[str, float]]:
    with _lock:
        return {k: {"calls": v[0], "total_s": round(v[1], 6), "mean_ms": round(v[1] / v[0] * 1e3, 4),
                    "max_ms": round(v[2] * 1e3, 4)} for k, v in sorted(_stages.items())}


def reset() -> None:
    with _lock:
        _stages.clear()
