"""GPU parity of the encoder kernels (through the C-ABI) vs the CPU oracle.

The device computes with fp16 weights/activations and fp32 accumulation; the oracle is float32
on the SAME fp16-rounded weights.  Tolerances (documented in DESIGN.md):
  single ops ........ |err| <= 2e-3 * scale (one fp16 rounding of the output + fp32 sum order)
  full forward ...... unit-norm embeddings: max |err| <= 4e-3 and cosine >= 0.9999
"""
import dataclasses
import os

import numpy as np
import pytest
import torch

from oracle import encoder_oracle as E

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def N():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from multimodal_rag_amd import _native

    _native.lib()
    return _native


def dev16(x):
    return torch.from_numpy(np.asarray(x, np.float32)).cuda().half().contiguous()


def dev32(x):
    return torch.from_numpy(np.asarray(x, np.float32)).cuda().contiguous()


def r16(x):
    return np.asarray(x, np.float32).astype(np.float16).astype(np.float32)


@pytest.mark.parametrize("M,K,Nf", [(70, 128, 128), (256, 384, 1152), (1000, 384, 384), (513, 1536, 384),
                                   (4096, 768, 2304), (3000, 768, 768), (2048, 3072, 768), (8192, 768, 3072)])
@pytest.mark.parametrize("act,with_resid", [(0, False), (1, False), (0, True), (2, False)])
def test_linear(N, M, K, Nf, act, with_resid):
    g = np.random.default_rng(M + K + Nf + act)
    x = r16(g.standard_normal((M, K)) * 0.5)
    w = r16(g.standard_normal((Nf, K)) * 0.05)
    b = (g.standard_normal(Nf) * 0.1).astype(np.float32)
    res = r16(g.standard_normal((M, Nf))) if with_resid else None
    y = x @ w.T + b
    if act == 1:
        y = E.gelu_erf(y)
    elif act == 2:
        y = E.quick_gelu(y)
    y = r16(y)  # the kernel rounds the activated value to fp16 before the residual add
    if with_resid:
        y = y + res
    out = N.linear_f16(dev16(x), dev16(w), dev32(b), act, dev16(res) if with_resid else None)
    torch.cuda.synchronize()
    got = out.float().cpu().numpy()
    scale = max(1.0, float(np.abs(y).max()))
    assert np.abs(got - y).max() <= 2e-3 * scale, float(np.abs(got - y).max())


@pytest.mark.parametrize("M,K,Nf", [(1, 384, 384), (7, 384, 1152), (16, 1536, 384), (32, 384, 1536), (33, 768, 2304),
                                   (64, 3072, 768), (5, 512, 100), (40, 64, 36), (64, 2048, 512)])
@pytest.mark.parametrize("act,with_resid", [(0, False), (1, False), (0, True), (2, True)])
def test_linear_few_tokens(N, M, K, Nf, act, with_resid):
    """M <= 64 takes the split-K kernel of the single-query path (ragged feature blocks, 1-2 token blocks)"""
    test_linear(N, M, K, Nf, act, with_resid)


def test_linear_few_tokens_exact_integers(N):
    g = np.random.default_rng(2)
    for M, K, Nf in ((3, 192, 68), (64, 448, 96), (33, 1536, 32)):
        x = g.integers(-4, 5, (M, K)).astype(np.float32)
        w = g.integers(-2, 3, (Nf, K)).astype(np.float32)
        w[:, 0] += np.arange(Nf) % 5
        out = N.linear_f16(dev16(x), dev16(w))
        assert np.array_equal(out.float().cpu().numpy(), x @ w.T), (M, K, Nf)


@pytest.mark.parametrize("M,K,Nf", [(22000, 768, 768), (8300, 128, 2304), (17000, 192, 1000), (65536, 64, 256)])
@pytest.mark.parametrize("act,with_resid", [(0, False), (1, False), (0, True), (2, True)])
def test_linear_one_tile_per_cu_and_more(N, M, K, Nf, act, with_resid):
    """>= 256 tiles of 256x256: the persistent kernel (ragged last token tile, ragged last feature tile, the two-slab
    minimum of its ring; K = 64 stays on the one-tile-per-workgroup kernel)"""
    test_linear(N, M, K, Nf, act, with_resid)


def test_linear_persistent_exact_integers(N):
    """every (tile, wave, lane, register) of the persistent kernel's direct store lands on its own element"""
    g = np.random.default_rng(5)
    M, K, Nf = 66000, 192, 768
    x = g.integers(-4, 5, (M, K)).astype(np.float32)
    w = g.integers(-2, 3, (Nf, K)).astype(np.float32)
    w[:, 0] += np.arange(Nf) % 7
    x[:, 1] += np.arange(M) % 3
    res = g.integers(-8, 9, (M, Nf)).astype(np.float32)
    out = N.linear_f16(dev16(x), dev16(w), dev32(np.arange(Nf) % 4), 0, dev16(res))
    assert np.array_equal(out.float().cpu().numpy(), x @ w.T + (np.arange(Nf) % 4) + res)


@pytest.fixture
def mfma32(N):
    """the persistent GEMM on v_mfma_f32_32x32x16_f16 (developer switch 8192: the A/B partner of the default 16x16x32 form)"""
    import ctypes

    L = N.lib()
    L.mmrag_internal_set_debug.argtypes = [ctypes.c_uint]
    L.mmrag_internal_set_debug(8192)
    yield
    L.mmrag_internal_set_debug(0)


@pytest.mark.parametrize("M,K,Nf", [(22000, 768, 768), (8300, 128, 2304), (17000, 192, 1000), (65536, 768, 2304)])
@pytest.mark.parametrize("act,with_resid", [(0, False), (1, False), (0, True), (2, True)])
def test_linear_persistent_32x32x16(N, mfma32, M, K, Nf, act, with_resid):
    test_linear(N, M, K, Nf, act, with_resid)


def test_linear_persistent_32x32x16_exact_integers(N, mfma32):
    """(the default 16x16x32 form -- W rows interleaved four ways at DMA time, 8-byte stores -- is what
    test_linear_persistent_exact_integers and test_linear_one_tile_per_cu_and_more run)"""
    test_linear_persistent_exact_integers(N)


def test_linear_exact_integers(N):
    """integer data is exact in fp16 x fp16 -> fp32: catches any fragment / epilogue index swap"""
    g = np.random.default_rng(0)
    M, K, Nf = 300, 192, 256
    x = g.integers(-4, 5, (M, K)).astype(np.float32)
    w = g.integers(-2, 3, (Nf, K)).astype(np.float32)
    w[:, 0] += np.arange(Nf) % 5
    out = N.linear_f16(dev16(x), dev16(w))
    assert np.array_equal(out.float().cpu().numpy(), x @ w.T)


@pytest.mark.parametrize("T,H", [(5, 128), (1000, 384), (777, 768), (64, 512)])
def test_layernorm(N, T, H):
    g = np.random.default_rng(T)
    x = r16(g.standard_normal((T, H)) * 2 + 0.3)
    gm = (1 + 0.1 * g.standard_normal(H)).astype(np.float32)
    bt = (0.1 * g.standard_normal(H)).astype(np.float32)
    got = N.layernorm_f16(dev16(x), dev32(gm), dev32(bt), 1e-12).float().cpu().numpy()
    ref = E.layer_norm(x, gm, bt, 1e-12)
    assert np.abs(got - ref).max() <= 4e-3


def test_embed_ln(N):
    g = np.random.default_rng(1)
    V, P, H, T = 500, 64, 384, 333
    tok, pos, typ = (r16(g.standard_normal((n, H)) * 0.1) for n in (V, P, 2))
    gm = (1 + 0.1 * g.standard_normal(H)).astype(np.float32)
    bt = (0.1 * g.standard_normal(H)).astype(np.float32)
    ids = g.integers(0, V, T).astype(np.int32)
    pid = g.integers(0, P, T).astype(np.int32)
    got = N.embed_ln_f16(torch.from_numpy(ids).cuda(), torch.from_numpy(pid).cuda(), dev16(tok), dev16(pos),
                         dev16(typ[0]), dev32(gm), dev32(bt), 1e-12).float().cpu().numpy()
    ref = E.layer_norm(tok[ids] + pos[pid] + typ[0], gm, bt, 1e-12)
    assert np.abs(got - ref).max() <= 4e-3


@pytest.mark.parametrize("H,heads,lens,causal", [
    (384, 12, [1, 7, 64, 65, 128, 129, 256], False),
    (768, 12, [300, 5, 512, 77], False),
    (128, 4, [33, 64], False),
    (512, 8, [77, 20, 77], True),
    (768, 12, [200, 140, 256], True),      # K / V resident in LDS (129-256 keys), causal
    (384, 12, [130, 255], True),
    (768, 12, [300, 400, 31], True),       # resident, up to 512 keys
    (768, 12, [129, 256, 200, 3], False),
])
def test_attention(N, H, heads, lens, causal):
    g = np.random.default_rng(H + len(lens))
    T = sum(lens)
    qkv = r16(g.standard_normal((T, 3 * H)))
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    got = N.attention_f16(dev16(qkv), torch.from_numpy(cu).cuda(), max(lens), heads, causal).float().cpu().numpy()
    ref = np.concatenate([E.attention(qkv[a:b, :H], qkv[a:b, H:2 * H], qkv[a:b, 2 * H:], heads, causal)
                          for a, b in zip(cu[:-1], cu[1:])])
    assert np.abs(got - ref).max() <= 3e-3, float(np.abs(got - ref).max())


def test_attention_spiked_scores_force_rescale(N):
    """one key dominates late in the sequence: exercises the online-softmax rescale branch"""
    H, heads, S = 128, 4, 200
    g = np.random.default_rng(4)
    qkv = r16(g.standard_normal((S, 3 * H)) * 0.3)
    qkv[150, H:2 * H] = r16(qkv[10, :H] * 40)  # key 150 aligned with query 10
    cu = np.array([0, S], np.int32)
    got = N.attention_f16(dev16(qkv), torch.from_numpy(cu).cuda(), S, heads).float().cpu().numpy()
    ref = E.attention(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], heads)
    assert np.abs(got - ref).max() <= 3e-3


@pytest.mark.parametrize("pool", [0, 1])
def test_pool_normalize(N, pool):
    g = np.random.default_rng(2)
    lens = [1, 9, 256, 40]
    H = 384
    x = r16(g.standard_normal((sum(lens), H)))
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    got = N.pool_normalize_f16(dev16(x), torch.from_numpy(cu).cuda(), pool).cpu().numpy()
    rows = [x[a:b].mean(0) if pool == 0 else x[a] for a, b in zip(cu[:-1], cu[1:])]
    ref = E.l2_normalize(np.stack(rows))
    assert np.abs(got - ref).max() <= 1e-5
    assert np.allclose(np.linalg.norm(got, axis=1), 1.0, atol=1e-5)


@pytest.mark.parametrize("name,shape", [("tiny", E.TINY), ("minilm", E.MINILM_L6), ("bge", E.BGE_BASE)])
@pytest.mark.parametrize("pool", ["mean", "cls"])
def test_full_forward_vs_oracle_and_transformers_golden(N, golden_dir, name, shape, pool):
    from multimodal_rag_amd.encoder import DeviceEncoder, EncoderConfig

    z = np.load(os.path.join(golden_dir, f"encoder_{name}.npz"))
    seed, ids, lens = int(z["seed"]), z["ids"], z["lens"]
    seqs = [ids[b, :n].tolist() for b, n in enumerate(lens)]
    w = E.make_bert_weights(shape, seed)
    cfg = EncoderConfig(name, shape.n_layers, shape.hidden, shape.n_heads, shape.intermediate, shape.vocab,
                        shape.max_pos, max_seq_length=shape.max_pos, pool=pool, ln_eps=shape.ln_eps)
    enc = DeviceEncoder(cfg, w, "cuda:0")
    got = enc.encode_ids(seqs).cpu().numpy()
    oracle16 = E.bert_encode(dataclasses.replace(shape, pool=pool), E.round_weights_fp16(w), seqs)
    golden = z[pool]  # transformers.BertModel, fp32 weights
    assert np.allclose(np.linalg.norm(got, axis=1), 1.0, atol=1e-4)
    err = np.abs(got - oracle16).max()
    cos = (got * oracle16).sum(1).min()
    assert err <= 4e-3 and cos >= 0.9999, (float(err), float(cos))
    # and the fp16 device result stays close to the float32 transformers output
    assert (got * golden).sum(1).min() >= 0.999


@pytest.mark.parametrize("shape", [E.MINILM_L6, E.BGE_BASE], ids=["minilm", "bge"])
@pytest.mark.parametrize("lens", [[9], [17, 5, 30], [64], [1], [33, 31]])
@pytest.mark.parametrize("pool", ["mean", "cls"])
def test_single_query_forward_vs_oracle(N, shape, lens, pool):
    """The online /query shape (api.py:338 is a batch of ONE; T <= 64 tokens) takes its own forward: LayerNorms folded
    into the neighbouring GEMMs, 16-feature workgroups, and for one sequence a HIP graph per token count
    (`DeviceEncoder.encode_one`).  Checked here against the float32 oracle DIRECTLY (round-2 verdict: it was only
    compared with the library's other forward): same fp16-rounded weights, LayerNorm gamma / beta well away from
    1 / 0, max |delta| <= 4e-3 and cosine >= 0.9999 like the long-sequence forward."""
    from multimodal_rag_amd.encoder import DeviceEncoder, EncoderConfig

    w = E.make_bert_weights(shape, seed=21)
    g = np.random.default_rng(22)
    for k in w:
        if k.endswith("LayerNorm.weight"):
            w[k] = (1.0 + 0.3 * g.standard_normal(w[k].shape)).astype(np.float32)
        elif k.endswith("LayerNorm.bias"):
            w[k] = (0.2 * g.standard_normal(w[k].shape)).astype(np.float32)
    cfg = EncoderConfig("q", shape.n_layers, shape.hidden, shape.n_heads, shape.intermediate, shape.vocab,
                        shape.max_pos, max_seq_length=shape.max_pos, pool=pool, ln_eps=shape.ln_eps)
    enc = DeviceEncoder(cfg, w, "cuda:0")
    seqs = [g.integers(1000, shape.vocab, n).tolist() for n in lens]
    want = E.bert_encode(dataclasses.replace(shape, pool=pool), E.round_weights_fp16(w), seqs)
    got = enc.encode_ids(seqs).cpu().numpy()
    err, cos = np.abs(got - want).max(), (got * want).sum(1).min()
    assert err <= 4e-3 and cos >= 0.9999, (float(err), float(cos))
    # one sequence at a time: the graph replay path (captured on the first call of a token count, replayed on the second)
    for i, sq in enumerate(seqs):
        for _ in range(2):
            one = enc.encode_one(sq).cpu().numpy()
            assert one.shape == (1, shape.hidden)
            e1, c1 = np.abs(one[0] - want[i]).max(), float((one[0] * want[i]).sum())
            assert e1 <= 4e-3 and c1 >= 0.9999, (i, float(e1), c1)
    if len(seqs) == 1:
        assert enc._graphs.get(len(seqs[0])) is not None     # the graph was captured, not the fallback


@pytest.mark.parametrize("hidden,heads,inter,lens", [(384, 12, 1536, [9]), (384, 12, 1536, [17, 5, 30]), (768, 12, 3072, [64]),
                                                      (768, 12, 3072, [3, 40]), (1024, 16, 2048, [33])])
def test_single_query_forward_without_layernorm_launches(N, hidden, heads, inter, lens):
    """T <= 64 (the online /query shape) runs BERT with the LayerNorms folded into the neighbouring GEMMs
    (csrc/encoder.hip "the single-query path without LayerNorm launches").  Same embeddings as the forward WITH the
    LayerNorm launches (developer switch 512), which the oracle / transformers goldens pin; non-trivial gamma / beta;
    one to three sequences, token blocks of 16 both full and ragged."""
    import ctypes

    from multimodal_rag_amd.encoder import DeviceEncoder, EncoderConfig, random_bert_weights

    cfg = EncoderConfig("t", 4, hidden, heads, inter, vocab=2000, max_pos=128, max_seq_length=128, pool="mean")
    w = random_bert_weights(cfg, seed=11, device="cuda:0", std=0.05)
    g = torch.Generator(device="cuda:0").manual_seed(5)
    for k in list(w):
        if "LayerNorm.weight" in k:
            w[k] = 1.0 + 0.3 * torch.randn(w[k].shape, device="cuda:0", generator=g)
        elif "LayerNorm.bias" in k:
            w[k] = 0.2 * torch.randn(w[k].shape, device="cuda:0", generator=g)
    enc = DeviceEncoder(cfg, w, "cuda:0")
    rng = np.random.default_rng(6)
    seqs = [rng.integers(5, 2000, n).tolist() for n in lens]
    L = N.lib()
    L.mmrag_internal_set_debug.argtypes = [ctypes.c_uint]
    try:
        L.mmrag_internal_set_debug(512)          # DBG_ENCODER_LN_PASSES
        ref = enc.encode_ids(seqs).cpu().numpy()
        L.mmrag_internal_set_debug(0)
        got = enc.encode_ids(seqs).cpu().numpy()
        again = enc.encode_ids(seqs).cpu().numpy()
    finally:
        L.mmrag_internal_set_debug(0)
    assert np.array_equal(got, again)
    assert np.abs(np.linalg.norm(got, axis=1) - 1).max() < 1e-3
    assert np.abs(got - ref).max() <= 1e-3, float(np.abs(got - ref).max())
