"""CPU, multi-process (gloo, world_size 2 and 4): the N>1 path of the sharded search -- shard
ranges, all-gather of per-rank top-k, host merge -- with the oracle standing in for the shard
kernel (the kernel itself is covered by the -m gpu tests).  Result must equal the single-shard
answer exactly, including cross-shard ties."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from multimodal_rag_amd.sharded import ShardedSearch, shard_range
from oracle import search_oracle as O

N_TOTAL, D, B, K = 1003, 48, 13, 5


def corpus_and_queries():
    g = np.random.default_rng(7)
    c = g.standard_normal((N_TOTAL, D)).astype(np.float32)
    c /= np.linalg.norm(c, axis=1, keepdims=True)
    c[900] = c[4]            # exact duplicate across shards -> tie broken by lower global row
    q = g.standard_normal((B, D)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    q[0] = c[4]
    return c, q


def _worker(rank, world, port, merge, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        c, q = corpus_and_queries()
        lo, hi = shard_range(N_TOTAL, world, rank)

        def local(slot, out_s, out_r):
            s, r = O.cosine_topk(q, c[lo:hi], K, row_offset=lo)
            out_s.copy_(torch.from_numpy(s))
            out_r.copy_(torch.from_numpy(r))

        ss = ShardedSearch(B, K, world, rank, torch.device("cpu"), local, merge=merge)
        ss.launch(0)
        ss.launch(1)          # two batches in flight, finished in order
        a = ss.finish(0)
        b = ss.finish(1)
        if rank == 0:
            np.savez(os.path.join(out_dir, f"out_{world}.npz"), s=a[0].numpy(), r=a[1].numpy(), s2=b[0].numpy())
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_search_matches_single_shard(world, tmp_path):
    mp.spawn(_worker, args=(world, _free_port(), "host", str(tmp_path)), nprocs=world, join=True)
    z = np.load(tmp_path / f"out_{world}.npz")
    c, q = corpus_and_queries()
    es, er = O.cosine_topk(q, c, K)
    assert np.array_equal(z["r"], er) and np.allclose(z["s"], es, atol=1e-6)
    assert list(z["r"][0][:2]) == [4, 900]
    assert np.array_equal(z["s"], z["s2"])


def test_shard_ranges_cover_exactly():
    for n in (0, 1, 7, 1000, 1_000_000):
        for w in (1, 2, 4, 8):
            parts = [shard_range(n, w, r) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            assert all(hi - lo <= -(-n // w) for lo, hi in parts)
