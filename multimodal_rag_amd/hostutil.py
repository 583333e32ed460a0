"""Small host-side pieces shared by the two mirrors of the reference interface (embedder.py, retriever.py)."""
from __future__ import annotations

import asyncio
import logging
from typing import Any, Awaitable, Callable, Dict, Optional


class CountingLRU:
    """Bounded least-recently-used map that counts hits and misses.

    Both caches of the reference are this object with different payloads (`LRUCache`, app/utils/embedder.py:26-80;
    `DocumentCache`, app/utils/retriever.py:35-90): same method names, same statistics keys, hit rate rounded to three
    places.  A plain dict keeps insertion order, so "touch" = take the entry out and put it back at the young end."""

    def __init__(self, maxsize: int):
        self.maxsize = maxsize
        self.cache: Dict[str, Any] = {}
        self.hits = 0
        self.misses = 0

    def get(self, key: str) -> Optional[Any]:
        try:
            value = self.cache.pop(key)
        except KeyError:
            self.misses += 1
            return None
        self.cache[key] = value
        self.hits += 1
        return value

    def put(self, key: str, value: Any) -> None:
        if self.cache.pop(key, None) is None and len(self.cache) >= self.maxsize:
            del self.cache[next(iter(self.cache))]      # the oldest entry
        self.cache[key] = value

    def invalidate(self, key: str) -> None:
        self.cache.pop(key, None)

    def clear(self) -> None:
        self.cache.clear()
        self.hits = self.misses = 0

    def get_stats(self) -> Dict[str, Any]:
        looked = self.hits + self.misses
        return {"size": len(self.cache), "maxsize": self.maxsize, "hits": self.hits, "misses": self.misses,
                "hit_rate": round(self.hits / looked, 3) if looked else 0.0}


async def call_with_retry(what: str, fn: Callable[..., Any], *args: Any, attempts: int = 3,
                          sleep: Callable[[float], Awaitable[Any]] = asyncio.sleep,
                          log: Optional[logging.Logger] = None, **kwargs: Any) -> Any:
    """Run the blocking engine call `fn(*args, **kwargs)` in a worker thread; on an exception wait 1 s, 2 s, 4 s ...
    and try again, `attempts` times in all, then let the last exception out.  This is the one retry policy the
    reference repeats around every store / query / delete call (embedder.py:514-537, :592-617, :628-656;
    retriever.py:264-309, :470-494)."""
    log = log or logging.getLogger(__name__)
    for n in range(attempts):
        try:
            return await asyncio.to_thread(fn, *args, **kwargs)
        except Exception as e:   # noqa: BLE001 -- the policy is "any exception"
            if n + 1 == attempts:
                log.error("%s failed after %d attempts: %s", what, attempts, e)
                raise
            log.warning("%s: attempt %d failed (%s); retrying in %d s", what, n + 1, e, 2 ** n)
            await sleep(2 ** n)


def load_hostrows():
    """the C extension that builds a batch's result lists (csrc/hostrows.c, built by `python -m multimodal_rag_amd.build`),
    or None when it has not been built: the callers keep their plain-Python form of the same loops"""
    import importlib.util
    import os
    import sysconfig

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib",
                        "_hostrows" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))
    if os.environ.get("MMRAG_NO_HOSTROWS") or not os.path.exists(path):
        return None
    try:
        spec = importlib.util.spec_from_file_location("_hostrows", path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    except Exception:   # noqa: BLE001 -- an unloadable helper must not take the service down
        return None
