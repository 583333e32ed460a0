"""GPU parity: libmmrag's fused cosine GEMM + top-k (through the C-ABI) vs the CPU oracle.

Bar: identical top-k id sets (candidates within 2e-4 of the k-th score
interchangeable), cosine within 1e-4; on exactly representable integer data: bit-exact.
"""
import numpy as np
import pytest
import torch

from oracle import search_oracle as O

pytestmark = pytest.mark.gpu

TOL = 1e-4  # north star: "cosine scores within 1e-4 fp32"


@pytest.fixture(scope="module")
def N():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from multimodal_rag_amd import _native

    _native.lib()
    return _native


def unit_rows(n, d, seed):
    g = np.random.default_rng(seed)
    x = g.standard_normal((n, d), dtype=np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x


def to_dev(N, x, dtype):
    """pad to the kernel's leading dimension, cast to storage dtype; also return the exact
    float32 view of the stored values for the oracle"""
    n, d = x.shape
    ld = N.padded_dim(d, dtype)
    t = torch.zeros((max(n, 1), ld), dtype=dtype, device="cuda")
    if n:
        t[:n, :d] = torch.from_numpy(x).to("cuda").to(dtype)
    stored = t[:n, :d].to(torch.float32).cpu().numpy()
    return t, stored


def run(N, q, c, k, dtype, row_offset=0, alive=None, dbg=0):
    qd, qs = to_dev(N, q, dtype)
    cd, cs = to_dev(N, c, dtype)
    bits = None
    if alive is not None:
        words = np.zeros((c.shape[0] + 31) // 32 + 8, dtype=np.uint32)
        idx = np.nonzero(alive)[0]
        np.bitwise_or.at(words, idx // 32, (np.uint32(1) << (idx % 32).astype(np.uint32)))
        bits = torch.from_numpy(words.view(np.int32)).to("cuda")
    s, r = N.cosine_topk(qd, cd, c.shape[0], c.shape[1], k, row_offset=row_offset, alive_bits=bits, dbg=dbg)
    torch.cuda.synchronize()
    es, er = O.cosine_topk(qs, cs, k, row_offset=row_offset, alive=alive)
    return s.cpu().numpy(), r.cpu().numpy(), es, er


def check(s, r, es, er):
    assert r.shape == er.shape and s.shape == es.shape
    fin = np.isfinite(es)
    assert np.array_equal(np.isfinite(s), fin)
    assert np.array_equal(r[~fin], er[~fin])  # -1 padding
    assert np.all(np.abs(s[fin] - es[fin]) <= TOL)
    assert np.all(np.diff(s, axis=1)[fin[:, 1:]] <= 0)  # descending
    assert O.same_topk_sets(r, s, er, es)


def test_wal70_golden(N, wal70):
    """real all-MiniLM-L6-v2 vectors from the reference's Chroma WAL: all-pairs top-5"""
    V = wal70["vectors"]
    s, r, es, er = run(N, V, V, 5, torch.float32)
    assert np.array_equal(r, wal70["top_rows"])
    assert np.all(np.abs(s - wal70["top_cos"]) <= 1e-5)
    assert np.array_equal(r, er)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("B,n,d,k", [
    (1, 1000, 384, 5), (33, 5003, 384, 5), (64, 777, 768, 3), (100, 9000, 768, 5),
    (256, 20000, 768, 5), (300, 4097, 512, 5), (7, 2500, 100, 10), (70, 3000, 384, 20), (5, 300, 64, 1),
])
def test_random_parity(N, dtype, B, n, d, k):
    q = unit_rows(B, d, 1)
    c = unit_rows(n, d, 2)
    check(*run(N, q, c, k, dtype))


@pytest.mark.parametrize("B,n,d,k", [(130, 6000, 384, 10), (257, 5000, 768, 20), (100, 2000, 256, 20),
                                      (200, 300_000, 128, 10), (256, 200_000, 64, 20)])
def test_deep_lists_large_batches(N, B, n, d, k):
    """k = 10 / 20 on the 128- and 256-query shapes; the two big cases take the sample pre-pass"""
    q = unit_rows(B, d, 11)
    c = unit_rows(n, d, 12)
    check(*run(N, q, c, k, torch.float16))


@pytest.mark.parametrize("B,n,d,k", [(520, 70_000, 128, 5), (1024, 200_000, 64, 5), (700, 200_000, 64, 10)])
def test_batches_above_256_share_corpus_tiles(N, B, n, d, k):
    """several query groups per corpus tile (co-scheduled on one XCD); the two big cases take the pre-pass"""
    q = unit_rows(B, d, 21)
    c = unit_rows(n, d, 22)
    check(*run(N, q, c, k, torch.float16))


def test_deep_lists_exact_integers(N):
    g = np.random.default_rng(6)
    n, d, B = 210_000, 64, 160
    c = g.integers(-3, 4, size=(n, d)).astype(np.float32)
    q = g.integers(-3, 4, size=(B, d)).astype(np.float32)
    for k in (10, 20):
        s, r, es, er = run(N, q, c, k, torch.float16)
        assert np.array_equal(s, es)
        assert np.array_equal(r, er)  # heavy ties across tiles, workgroups and the sample pre-pass


def test_exact_integers_catch_layout_swaps(N):
    """small-integer data is exact in fp16/fp32: any fragment/row-map mistake changes ids"""
    g = np.random.default_rng(5)
    n, d, B = 1500, 128, 96
    c = g.integers(-3, 4, size=(n, d)).astype(np.float32)
    q = g.integers(-3, 4, size=(B, d)).astype(np.float32)
    c[:, 0] += np.arange(n) % 7  # asymmetric
    for dtype in (torch.float16, torch.float32):
        s, r, es, er = run(N, q, c, 5, dtype)
        assert np.array_equal(s, es)
        assert np.array_equal(r, er)  # ties are frequent here: exercises lower-row-first


def test_ties_duplicates_lower_row_first(N):
    c = unit_rows(600, 384, 3)
    c[300:600] = c[0:300]  # every row twice
    q = c[[5, 17, 299, 0]]
    s, r, es, er = run(N, q, c, 5, torch.float32)
    assert np.array_equal(r, er)
    assert list(r[0][:2]) == [5, 305]


def test_ragged_small_and_empty(N):
    q = unit_rows(4, 384, 1)
    for n in (0, 1, 3, 5, 255, 256, 257):
        c = unit_rows(n, 384, 2) if n else np.zeros((0, 384), np.float32)
        s, r, es, er = run(N, q, c, 5, torch.float16)
        check(s, r, es, er)
        assert np.array_equal(r, er)


def test_alive_mask_and_row_offset(N):
    q = unit_rows(40, 768, 1)
    c = unit_rows(5000, 768, 2)
    alive = np.random.default_rng(9).random(5000) > 0.3
    alive[:700] = False
    s, r, es, er = run(N, q, c, 5, torch.float16, row_offset=1_000_000_007, alive=alive)
    check(s, r, es, er)
    assert np.all(alive[r - 1_000_000_007])


def test_planted_neighbours(N):
    """queries = corpus rows + small noise: the known answer is that row"""
    c = unit_rows(30000, 768, 2)
    rows = np.arange(0, 30000, 117)[:256]
    q = c[rows] + 0.02 * unit_rows(256, 768, 7)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    s, r, es, er = run(N, q.astype(np.float32), c, 5, torch.float16)
    assert np.array_equal(r[:, 0], rows)
    check(s, r, es, er)


def test_merge_topk_device_and_host(N):
    g = np.random.default_rng(3)
    G, B, k = 8, 37, 5
    s = g.standard_normal((G, B, k)).astype(np.float32)
    s = -np.sort(-s, axis=2)
    r = g.permutation(G * B * k).reshape(G, B, k).astype(np.int64)
    s[3, :, 3:] = -np.inf
    r[3, :, 3:] = -1
    s[1, 0, 0] = s[2, 0, 0] = 9.0  # tie across shards -> lower global row wins
    es, er = O.merge_topk(s, r, k)
    ds, dr = N.merge_topk(torch.from_numpy(s).cuda(), torch.from_numpy(r).cuda(), k)
    hs, hr = N.merge_topk_host(torch.from_numpy(s), torch.from_numpy(r), k)
    assert np.array_equal(ds.cpu().numpy(), es) and np.array_equal(dr.cpu().numpy(), er)
    assert np.array_equal(hs.numpy(), es) and np.array_equal(hr.numpy(), er)


def test_bad_arguments_fail_loudly(N):
    q = torch.zeros((4, 384), dtype=torch.float32, device="cuda")
    c = torch.zeros((10, 384), dtype=torch.float32, device="cuda")
    with pytest.raises(N.MMRagNativeError):
        N.cosine_topk(q, c, 10, 384, 21)
    with pytest.raises(N.MMRagNativeError):
        N.cosine_topk(q, c, 11, 384, 5)
    with pytest.raises(N.MMRagNativeError):
        N.cosine_topk(q.cpu(), c.cpu(), 10, 384, 5)


def test_randomised_shapes_against_oracle(N):
    """60 random (B, n, d, k, dtype, mask, offset) draws, biased towards the kernel's seams: batch sizes around the
    64/128/256-query shapes and their multiples, row counts around tile (256) and pre-pass (3 tiles per CU)
    boundaries, dims around the 128-byte slab, all three list depths, sparse and dense alive masks."""
    g = np.random.default_rng(2024)
    seams_b = [1, 2, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 300, 511, 513, 640]
    seams_n = [1, 4, 5, 6, 255, 256, 257, 511, 513, 1000, 4099, 65_537, 196_607, 196_609, 230_000]
    for trial in range(60):
        B = int(g.choice(seams_b))
        n = int(g.choice(seams_n))
        d = int(g.choice([8, 33, 64, 96, 100, 128, 384]))
        if n > 100_000:
            d = min(d, 64)                      # keep the oracle's sgemm quick
            B = min(B, 257)
        k = int(g.choice([1, 3, 5, 6, 10, 17, 20]))
        dtype = [torch.float16, torch.float32, torch.bfloat16][int(g.integers(3))]
        alive = None
        mode = int(g.integers(4))
        if mode == 1:
            alive = g.random(n) > 0.5
        elif mode == 2:
            alive = np.zeros(n, bool)
            alive[g.integers(0, n, size=min(n, 7))] = True     # fewer live rows than k is possible
        off = int(g.choice([0, 12345, 1 << 33]))
        q = unit_rows(B, d, 1000 + trial)
        c = unit_rows(n, d, 2000 + trial)
        s, r, es, er = run(N, q, c, k, dtype, row_offset=off, alive=alive)
        try:
            check(s, r, es, er)
            if alive is not None:
                live = r[r >= 0] - off
                assert np.all(alive[live])
        except AssertionError as e:
            raise AssertionError(f"trial {trial}: B={B} n={n} d={d} k={k} {dtype} mask={mode} off={off}") from e


# ---- query-stationary kernel (csrc/search_qs.hip): more than 128 queries, fp16/bf16 rows of 768 / 1024 / 1536 bytes.
# The product sends shards of >= 6 x 256 rows per CU there (3.75 x 256 for the single-launch walk at list depth 5); DBG_FORCE_QS makes it take the small shards of these tests
# (>= 256 rows per CU).  Every case is also run through the slab-ring kernel (DBG_NO_QS): bit-for-bit agreement.
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("B,n,d", [
    (129, 70_000, 768), (256, 66_000, 768), (200, 65_537, 512), (256, 70_001, 384), (300, 131_123, 768),
    (1000, 140_000, 384), (256, 300_000, 768),   # the last one takes the sample pre-pass
])
def test_query_stationary_parity(N, dtype, B, n, d):
    q = unit_rows(B, d, 31)
    c = unit_rows(n, d, 32)
    s, r, es, er = run(N, q, c, 5, dtype, dbg=N.DBG_FORCE_QS)
    check(s, r, es, er)
    qd, _ = to_dev(N, q, dtype)
    cd, _ = to_dev(N, c, dtype)
    for dbg in (N.DBG_NO_QS, N.DBG_FORCE_QS | N.DBG_NO_PREPASS, N.DBG_NO_QS | N.DBG_NO_PREPASS, 0):
        s2, r2 = N.cosine_topk(qd, cd, n, d, 5, dbg=dbg)
        assert np.array_equal(r2.cpu().numpy(), r) and np.array_equal(s2.cpu().numpy(), s), dbg


@pytest.mark.parametrize("k", [10, 20])
@pytest.mark.parametrize("B,n,d,dtype", [(256, 70_001, 768, torch.float16), (300, 131_123, 384, torch.float16),
                                         (129, 300_000, 512, torch.float16), (600, 90_000, 768, torch.bfloat16)])
def test_query_stationary_deeper_lists(N, k, B, n, d, dtype):
    """k = 10 runs on the query-stationary kernel too (lists of that depth in LDS, a ring that gives up stages for them);
    k = 20 (api.py:163 caps top_k at 20) stays on the slab-ring kernel, DBG_FORCE_QS or not: same checks for both, with
    and without the sample pass, bit for bit between the kernels"""
    q = unit_rows(B, d, 61)
    c = unit_rows(n, d, 62)
    s, r, es, er = run(N, q, c, k, dtype, dbg=N.DBG_FORCE_QS)
    check(s, r, es, er)
    qd, _ = to_dev(N, q, dtype)
    cd, _ = to_dev(N, c, dtype)
    for dbg in (N.DBG_NO_QS, N.DBG_FORCE_QS | N.DBG_NO_PREPASS, 0):
        s2, r2 = N.cosine_topk(qd, cd, n, d, k, dbg=dbg)
        assert np.array_equal(r2.cpu().numpy(), r) and np.array_equal(s2.cpu().numpy(), s), dbg
    g = np.random.default_rng(63)
    alive = g.random(n) < 0.3
    check(*run(N, q, c, k, dtype, row_offset=123_456, alive=alive, dbg=N.DBG_FORCE_QS))


def test_query_stationary_exact_integers_and_ties(N):
    """integer data is exact in fp16: ids must match the oracle bit for bit, ties (many) -> lower row"""
    g = np.random.default_rng(16)
    n, d, B = 250_000, 384, 256
    c = g.integers(-2, 3, size=(n, d)).astype(np.float32)
    q = g.integers(-2, 3, size=(B, d)).astype(np.float32)
    for k in (1, 5, 10, 20):
        s, r, es, er = run(N, q, c, k, torch.float16, dbg=N.DBG_FORCE_QS)
        assert np.array_equal(s, es)
        assert np.array_equal(r, er)
        s, r, es, er = run(N, q, c, k, torch.float16, dbg=N.DBG_FORCE_QS | N.DBG_NO_PREPASS)   # no sample pass
        assert np.array_equal(s, es)
        assert np.array_equal(r, er)


def test_query_stationary_masks_offsets_ragged(N):
    """alive bitmask (tombstones / where filters), shard row offset, ragged last tile, fewer live rows than k"""
    n, d, B = 66_003, 768, 160
    q = unit_rows(B, d, 41)
    c = unit_rows(n, d, 42)
    g = np.random.default_rng(43)
    alive = g.random(n) < 0.5
    for dbg in (N.DBG_FORCE_QS, N.DBG_FORCE_QS | N.DBG_NO_PREPASS):
        check(*run(N, q, c, 5, torch.float16, row_offset=1_000_000, alive=alive, dbg=dbg))
    few = np.zeros(n, dtype=bool)
    few[[5, 40_000, n - 1]] = True            # 3 live rows < k
    none = np.zeros(n, dtype=bool)
    for dbg in (N.DBG_FORCE_QS, N.DBG_FORCE_QS | N.DBG_NO_PREPASS):
        check(*run(N, q, c, 5, torch.float16, alive=few, dbg=dbg))
        check(*run(N, q, c, 5, torch.float16, alive=none, dbg=dbg))


# ---- single-launch walk kernel (csrc/search_qsw.hip): list depth 5 on shards long enough for its in-kernel threshold
# exchange (>= 12 tiles of 64 rows per workgroup) and its ticketed tail (>= 24, one query group).  Every case: the
# default plan against the oracle, and bit for bit against the same kernel without tickets / without the exchange, the
# three-launch kernel of search_qs.hip and the slab-ring kernel.
@pytest.mark.parametrize("B,n,d,dtype", [
    (256, 400_003, 768, torch.float16),     # exchange + tickets, ragged last tile
    (300, 250_000, 768, torch.bfloat16),    # two query groups (128 walkers each): exchange, static tiles
    (1000, 200_000, 384, torch.float16),    # four groups of 64 walkers: four thresholds per workgroup; 768-byte rows
    (256, 420_000, 512, torch.float16),     # 1024-byte rows
    (130, 393_300, 768, torch.float16),     # a half-empty query group
])
def test_walk_kernel_parity(N, B, n, d, dtype):
    q = unit_rows(B, d, 71)
    c = unit_rows(n, d, 72)
    s, r, es, er = run(N, q, c, 5, dtype, dbg=N.DBG_FORCE_QS)
    check(s, r, es, er)
    qd, _ = to_dev(N, q, dtype)
    cd, _ = to_dev(N, c, dtype)
    for dbg in (N.DBG_FORCE_QS | N.DBG_NO_DYN, N.DBG_FORCE_QS | N.DBG_NO_SEED, N.DBG_FORCE_QS | N.DBG_NO_DYN | N.DBG_NO_SEED,
                N.DBG_FORCE_QS | N.DBG_OLD_QS, N.DBG_NO_QS, 0):
        s2, r2 = N.cosine_topk(qd, cd, n, d, 5, dbg=dbg)
        assert np.array_equal(r2.cpu().numpy(), r) and np.array_equal(s2.cpu().numpy(), s), hex(dbg)
    # the same workspace again: the exchange block of the last launch must not leak into this one
    ws = torch.empty(N.cosine_topk_workspace_bytes(B, n, 5), dtype=torch.uint8, device="cuda")
    q2d, _ = to_dev(N, unit_rows(B, d, 73), dtype)
    a = N.cosine_topk(q2d, cd, n, d, 5, workspace=ws)
    b = N.cosine_topk(qd, cd, n, d, 5, workspace=ws)
    assert np.array_equal(b[1].cpu().numpy(), r) and np.array_equal(b[0].cpu().numpy(), s)
    assert not np.array_equal(a[1].cpu().numpy(), r)


def test_walk_kernel_masks_offsets_and_exact_ties(N):
    """alive bitmask + shard row offset through the exchange and the ticketed tail; integer data (exact in fp16, ties
    everywhere -> lower row) bit for bit against the oracle; k < 5 takes the same lists"""
    n, d, B = 400_003, 768, 200
    q = unit_rows(B, d, 81)
    c = unit_rows(n, d, 82)
    g = np.random.default_rng(83)
    alive = g.random(n) < 0.5
    check(*run(N, q, c, 5, torch.float16, row_offset=3_000_000, alive=alive, dbg=N.DBG_FORCE_QS))
    few = np.zeros(n, dtype=bool)
    few[[7, 200_000, n - 1]] = True            # 3 live rows < k: the exchange never yields a threshold
    check(*run(N, q, c, 5, torch.float16, alive=few, dbg=N.DBG_FORCE_QS))
    ci = g.integers(-2, 3, size=(n, 384)).astype(np.float32)
    qi = g.integers(-2, 3, size=(256, 384)).astype(np.float32)
    for k in (1, 3, 5):
        s, r, es, er = run(N, qi, ci, k, torch.float16, dbg=N.DBG_FORCE_QS)
        assert np.array_equal(s, es) and np.array_equal(r, er)


def test_walk_kernel_ticketed_tail_is_complete_on_short_rows(N):
    """384-d rows make a tile only a dozen ring pieces: the ticket of a tail tile has to be waited for explicitly (the
    ring's hand-over waits do not cover it) -- without that wait one launch in a few lost a whole tail tile.  Integer
    data (exact scores, ties everywhere), the oracle once, then the same answer from every one of a dozen launches."""
    n, d, B = 400_003, 384, 256   # 24.4 tiles per workgroup: the last three positions are handed out by ticket
    g = np.random.default_rng(91)
    ci = g.integers(-2, 3, size=(n, d)).astype(np.float32)
    qi = g.integers(-2, 3, size=(B, d)).astype(np.float32)
    s, r, es, er = run(N, qi, ci, 5, torch.float16, dbg=N.DBG_FORCE_QS)
    assert np.array_equal(s, es) and np.array_equal(r, er)
    qd, _ = to_dev(N, qi, torch.float16)
    cd, _ = to_dev(N, ci, torch.float16)
    for launch in range(12):
        s2, r2 = N.cosine_topk(qd, cd, n, d, 5, dbg=N.DBG_FORCE_QS)
        assert np.array_equal(r2.cpu().numpy(), r) and np.array_equal(s2.cpu().numpy(), s), launch


# ---- fp32 storage, more than 64 queries: scores come from a 3-term bf16 split (csrc/search.hip); the bound is the
# north star's 1e-4, tested on the reference's own vectors and on random data; integer data stays bit-exact
def test_fp32_split_on_the_wal70_vectors(N, wal70):
    V = wal70["vectors"]
    g = np.random.default_rng(3)
    q = np.concatenate([V, V[g.integers(0, 70, 130)] + 0.05 * unit_rows(130, V.shape[1], 8)])
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    s, r, es, er = run(N, q.astype(np.float32), V, 5, torch.float32)          # B = 200 -> split path
    assert np.abs(s - es).max() <= 1e-4 and O.same_topk_sets(r, s, er, es)
    assert np.array_equal(r[:70, 0], np.arange(70))                            # every stored vector finds itself first
    s1, r1, _, _ = run(N, q[:60].astype(np.float32), V, 5, torch.float32)      # B = 60 -> exact f32 MFMA
    assert np.abs(s1 - s[:60]).max() <= 5e-5


@pytest.mark.parametrize("B,n,d", [(256, 100_000, 384), (200, 30_000, 768), (300, 70_001, 100)])
def test_fp32_split_random_and_exact_integers(N, B, n, d):
    q, c = unit_rows(B, d, 51), unit_rows(n, d, 52)
    s, r, es, er = run(N, q, c, 5, torch.float32)
    assert np.abs(s - es).max() <= 1e-4 and O.same_topk_sets(r, s, er, es)
    g = np.random.default_rng(53)
    ci = g.integers(-3, 4, size=(n, d)).astype(np.float32)
    qi = g.integers(-3, 4, size=(B, d)).astype(np.float32)
    s, r, es, er = run(N, qi, ci, 5, torch.float32)
    assert np.array_equal(s, es) and np.array_equal(r, er)                     # small integers: hi term exact, lo = 0


def test_fp32_scores_do_not_depend_on_the_batch_when_asked(N):
    """a float32 collection scores batches of more than 64 queries from a bf16 split (<= 4e-5 off), smaller
    ones exactly, so a query's score depended on how many requests the dispatcher batched with it.
    MMRAG_F32_EXACT_SEARCH (VectorIndex.f32_exact) keeps the exact float32 path for every batch size: a query's row of
    the result is bit for bit the same alone, among 65 and among 256.  Default mode: ids still equal on data whose
    score gaps exceed the split's error (the tie rule -- lower row first -- is the same on both paths)."""
    from multimodal_rag_amd.index import VectorIndex

    d, n = 384, 60_000
    c = unit_rows(n, d, 91)
    c[1000] = c[7]                      # an exact tie: row 7 must come first on every path
    q = unit_rows(256, d, 92)
    q[0] = c[7]
    ix = VectorIndex(d, dtype=torch.float32, device="cuda:0", capacity=n)
    ix.add_rows_device(torch.from_numpy(c).cuda(), None, None, [f"doc_a_text_{i}" for i in range(n)])
    qd = torch.from_numpy(q).cuda()
    ix.f32_exact = True
    s1, r1 = ix.search(qd[:1], 5)
    s65, r65 = ix.search(qd[:65], 5)
    s256, r256 = ix.search(qd, 5)
    assert torch.equal(s1, s65[:1]) and torch.equal(r1, r65[:1])
    assert torch.equal(s65, s256[:65]) and torch.equal(r65, r256[:65])
    assert r256[0, :2].tolist() == [7, 1000]
    es, er = O.cosine_topk(q, c, 5)
    assert np.abs(s256.cpu().numpy() - es).max() <= 2e-6
    ix.f32_exact = False                # the default: the split path above 64 queries
    s, r = ix.search(qd, 5)
    assert np.abs(s.cpu().numpy() - s256.cpu().numpy()).max() <= 4e-5
    gaps = np.abs(np.diff(es, axis=1)).min(axis=1)
    clear = np.nonzero(gaps > 2e-4)[0]  # queries whose top-5 scores are further apart than the split's error
    assert clear.size > 100 and np.array_equal(r.cpu().numpy()[clear], r256.cpu().numpy()[clear])
    assert r[0, :2].tolist() == [7, 1000]
