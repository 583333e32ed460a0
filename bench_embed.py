"""Embed leg of bench.py: chunks embedded / s (encoder forward + append into the corpus shard).
Lives next to bench.py (it times the oracle as the CPU baseline, so it is not part of the package).

Workload: bge-base-en-v1.5 architecture (BASELINE.json configs[2]; L=12, H=768, 12 heads,
I=3072, CLS pooling), random-init fp16 weights, synthetic token ids ([CLS] body [SEP]),
256 chunks x 256 tokens per step per GPU (CHUNK_SIZE=1000 characters ~ 250 word pieces).
Data-parallel: every rank embeds its own chunks into its own shard, no collective.
"""
from __future__ import annotations

import time

import numpy as np
import torch

from multimodal_rag_amd import _native
from multimodal_rag_amd.encoder import PRESETS, DeviceEncoder, random_bert_weights

CHUNKS_PER_STEP = 256
SEQ = 256
MFMA_F16_PEAK_TFLOPS = 2500.0


def synthetic_ids(n_chunks: int, seq: int, vocab: int, seed: int):
    g = np.random.default_rng(seed)
    ids = g.integers(1000, vocab, size=(n_chunks, seq), dtype=np.int64).astype(np.int32)
    ids[:, 0] = 101
    ids[:, -1] = 102
    return ids


def run(dev, rank: int, world: int, steps: int = 10, warmup: int = 2, with_cpu_baseline: bool = True,
        extras: bool = True):
    """`extras=False`: the headline shape only (tools/embed_once.py under rocprofv3: the kernel statistics of that run
    must be the per-layer table of the bge S=256 step and nothing else)."""
    import torch.distributed as dist

    cfg = PRESETS["BAAI/bge-base-en-v1.5"]
    w = random_bert_weights(cfg, seed=4321, device=dev)
    enc = DeviceEncoder(cfg, w, dev)
    ids_np = synthetic_ids(CHUNKS_PER_STEP, SEQ, cfg.vocab, seed=100 + rank)
    ids = torch.from_numpy(ids_np.reshape(-1)).to(dev)
    pos = torch.arange(SEQ, dtype=torch.int32, device=dev).repeat(CHUNKS_PER_STEP)
    cu = torch.arange(0, (CHUNKS_PER_STEP + 1) * SEQ, SEQ, dtype=torch.int32, device=dev)
    out = torch.empty((CHUNKS_PER_STEP, cfg.dim), dtype=torch.float32, device=dev)
    ld = _native.padded_dim(cfg.dim, torch.float16)
    shard = torch.zeros((CHUNKS_PER_STEP * (steps + warmup), ld), dtype=torch.float16, device=dev)

    def step(i):
        enc.forward_packed(ids, pos, cu, SEQ, out=out)
        _native.append_rows(shard, i * CHUNKS_PER_STEP, out, cfg.dim)

    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    t_warm = time.perf_counter() + 0.5   # clocks settle over ~100 ms of load: warm up by time, not by step count
    while time.perf_counter() < t_warm:
        step(0)
        torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for i in range(steps):
        step(warmup + i)
    e1.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank != 0:
        return None
    dev_ms = e0.elapsed_time(e1) / steps
    flops_step = CHUNKS_PER_STEP * enc.flops_per_sequence(SEQ)
    tflops = flops_step / (dev_ms * 1e-3) / 1e12
    res = {
        "metric": "chunks embedded/sec", "value": round(world * CHUNKS_PER_STEP * steps / dt, 1), "unit": "chunks/s",
        "steps": steps, "ms_per_step": round(dt / steps * 1e3, 3), "scaling": "weak", "dtype": "f16",
        "config": {"workload": f"embed: bge-base-en-v1.5 shape (L12 H768 I3072, CLS pool), random-init fp16, "
                               f"{CHUNKS_PER_STEP} chunks x {SEQ} tokens per step per GPU, forward + append to shard",
                   "chunks_per_step_per_gpu": CHUNKS_PER_STEP, "seq_len": SEQ, "parallelism": f"dp{world}"},
        "roofline": {"bound": "mfma", "kernel": "linear_persistent_kernel (+ attention_kernel)",
                     "achieved": round(tflops, 1), "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(tflops / MFMA_F16_PEAK_TFLOPS, 4), "traffic": None,
                     "flops_per_chunk": enc.flops_per_sequence(SEQ), "device_ms_per_step": round(dev_ms, 3)},
    }
    if with_cpu_baseline and world == 1:
        from oracle import encoder_oracle as E

        shape = E.BertShape(cfg.n_layers, cfg.hidden, cfg.n_heads, cfg.intermediate, cfg.vocab, cfg.max_pos,
                            cfg.ln_eps, cfg.pool)
        w16 = E.round_weights_fp16({k: v.float().cpu().numpy() for k, v in w.items()})
        n_cpu = 4
        seqs = [ids_np[i].tolist() for i in range(n_cpu)]
        t0 = time.perf_counter()
        ref = E.bert_encode(shape, w16, seqs)
        cdt = time.perf_counter() - t0
        got = out[:n_cpu].cpu().numpy()
        res["cpu_baseline"] = {"value": round(n_cpu / cdt, 3), "unit": "chunks/s", "cores": torch.get_num_threads(),
                               "kind": "port",
                               "sample": f"oracle/encoder_oracle.py (numpy fp32) on {n_cpu} chunks x {SEQ} tokens"}
        try:   # a fairer CPU figure: the torch module sentence-transformers wraps, at the reference's batch size
            res["cpu_baseline"] = cpu_baseline_torch(cfg, ids_np) or res["cpu_baseline"]
        except Exception as e:  # transformers missing or too old: keep the numpy figure
            res["cpu_baseline"]["torch_cpu_error"] = str(e)[:200]
        res["parity_vs_oracle"] = bool(np.abs(got - ref).max() <= 4e-3 and (got * ref).sum(1).min() >= 0.9999)
        res["max_abs_err_vs_oracle"] = float(np.abs(got - ref).max())
    if not extras:
        return res
    try:   # "bge 512 and 256", MiniLM (configs 1-2) 256, and ragged lengths
        res["other_shapes"] = bench_other_shapes(dev)
    except Exception as e:
        res["other_shapes"] = {"error": str(e)[:200]}
    try:
        res["clip_vit_b32"] = bench_clip_images(dev)
    except Exception as e:  # the headline numbers must survive a failure of the extra leg
        res["clip_vit_b32"] = {"error": str(e)}
    try:
        res["from_text"] = bench_from_text(enc, shard)
    except Exception as e:
        res["from_text"] = {"error": str(e)}
    return res


def _time_forward(enc, ids, pos, cu, max_len, out, seconds=0.4):
    for _ in range(2):
        enc.forward_packed(ids, pos, cu, max_len, out=out)
    torch.cuda.synchronize()
    t_end = time.perf_counter() + seconds / 2
    while time.perf_counter() < t_end:
        enc.forward_packed(ids, pos, cu, max_len, out=out)
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 0
    e0.record()
    t_end = time.perf_counter() + seconds / 2
    while time.perf_counter() < t_end:
        enc.forward_packed(ids, pos, cu, max_len, out=out)
        n += 1
        if n % 4 == 0:
            torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def bench_other_shapes(dev):
    """Forward-only chunks/s at the other shapes SURVEY 8d names: bge-base S=512, all-MiniLM-L6-v2 S=256, and
    bge-base with clipped log-normal lengths (mean ~180, max 256; no padded tokens are computed)."""
    res = {}
    F32_MFMA_PEAK_TFLOPS = 157.3   # v_mfma_f32_32x32x2_f32: the float32 matrix rate (MI355X_MICROARCH.md)
    for name, key, S, chunks, precision in (
            ("bge_base_s512", "BAAI/bge-base-en-v1.5", 512, 128, "fp16"),
            ("minilm_l6_s256", "sentence-transformers/all-MiniLM-L6-v2", 256, 256, "fp16"),
            # the reference's own precision (SentenceTransformer.encode is float32): csrc/encoder_f32.hip
            ("minilm_l6_s256_fp32", "sentence-transformers/all-MiniLM-L6-v2", 256, 256, "fp32"),
            ("bge_base_s256_fp32", "BAAI/bge-base-en-v1.5", 256, 64, "fp32")):
        cfg = PRESETS[key]
        enc = DeviceEncoder(cfg, random_bert_weights(cfg, seed=77, device=dev), dev, precision=precision)
        ids = torch.from_numpy(synthetic_ids(chunks, S, cfg.vocab, seed=9).reshape(-1)).to(dev)
        pos = torch.arange(S, dtype=torch.int32, device=dev).repeat(chunks)
        cu = torch.arange(0, (chunks + 1) * S, S, dtype=torch.int32, device=dev)
        out = torch.empty((chunks, cfg.dim), dtype=torch.float32, device=dev)
        ms = _time_forward(enc, ids, pos, cu, S, out)
        tf = chunks * enc.flops_per_sequence(S) / (ms * 1e-3) / 1e12
        peak = MFMA_F16_PEAK_TFLOPS if precision == "fp16" else F32_MFMA_PEAK_TFLOPS
        res[name] = {"chunks_per_s": round(chunks / ms * 1e3, 1), "ms_per_step": round(ms, 3), "chunks_per_step": chunks,
                     "seq_len": S, "dtype": "f16" if precision == "fp16" else "f32", "tflops": round(tf, 1),
                     "mfma_peak_tflops": peak, "mfma_frac": round(tf / peak, 4)}
        del enc
    cfg = PRESETS["BAAI/bge-base-en-v1.5"]
    enc = DeviceEncoder(cfg, random_bert_weights(cfg, seed=78, device=dev), dev)
    g = np.random.default_rng(12)
    lens = np.clip(np.exp(g.normal(np.log(170.0), 0.45, size=256)).astype(np.int64), 8, 256)
    T = int(lens.sum())
    ids = torch.from_numpy(g.integers(1000, cfg.vocab, size=T).astype(np.int32)).to(dev)
    pos = torch.from_numpy(np.concatenate([np.arange(n, dtype=np.int32) for n in lens])).to(dev)
    cu = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)).to(dev)
    out = torch.empty((256, cfg.dim), dtype=torch.float32, device=dev)
    ms = _time_forward(enc, ids, pos, cu, int(lens.max()), out)
    useful = float(sum(enc.flops_per_sequence(int(n)) for n in lens))
    res["bge_base_lognormal"] = {"chunks_per_s": round(256 / ms * 1e3, 1), "ms_per_step": round(ms, 3),
                                 "mean_tokens_per_chunk": round(T / 256, 1), "max_tokens": int(lens.max()),
                                 "useful_tflops": round(useful / (ms * 1e-3) / 1e12, 1)}
    return res


def cpu_baseline_torch(cfg, ids_np, batch: int = 32):
    """CPU baseline of the embed leg: `transformers.BertModel` (what sentence-transformers 2.2.2 wraps; that package
    itself is absent) with the same architecture, random weights, fp32, `batch` chunks x SEQ tokens -- the reference's
    batch size (api.py:93) -- then CLS/mean pooling + L2 normalise.  Throughput only; parity is checked against the
    numpy oracle on the GPU's own weights."""
    import transformers

    c = transformers.BertConfig(vocab_size=cfg.vocab, hidden_size=cfg.hidden, num_hidden_layers=cfg.n_layers,
                                num_attention_heads=cfg.n_heads, intermediate_size=cfg.intermediate,
                                max_position_embeddings=cfg.max_pos, layer_norm_eps=cfg.ln_eps)
    model = transformers.BertModel(c, add_pooling_layer=False).eval()
    x = torch.from_numpy(ids_np[:batch].astype(np.int64))
    best, best_threads = None, torch.get_num_threads()
    default_threads = torch.get_num_threads()
    # a GPU box hands one job a share of its cores (16 per GPU): the default thread count (all cores of the host)
    # oversubscribes that share, so try the share too and report the better one with ITS thread count
    try:
        with torch.no_grad():
            for threads in sorted({default_threads, min(default_threads, 16)}):
                torch.set_num_threads(threads)
                for _ in range(2):
                    t0 = time.perf_counter()
                    h = model(input_ids=x).last_hidden_state
                    e = h[:, 0] if cfg.pool == "cls" else h.mean(1)
                    e = torch.nn.functional.normalize(e, dim=1)
                    dt = time.perf_counter() - t0
                    if best is None or dt < best:
                        best, best_threads = dt, threads
    finally:
        torch.set_num_threads(default_threads)
    return {"value": round(batch / best, 2), "unit": "chunks/s", "cores": best_threads, "kind": "port",
            "sample": f"transformers.BertModel fp32 on CPU (the module sentence-transformers wraps), {batch} chunks x "
                      f"{x.shape[1]} tokens (reference batch size, api.py:93), best of 2 at 16 and at all threads"}


def synthetic_vocab(n: int = 30522, seed: int = 11):
    """A WordPiece-shaped vocabulary (specials at BERT's ids, whole words and ## pieces of 2-7 letters) and ~1000-char
    chunk texts drawn from it -- stands in for vocab.txt, which cannot be fetched here."""
    g = np.random.default_rng(seed)
    letters = np.array(list("abcdefghijklmnopqrstuvwxyz"))
    vocab = {}
    for i in range(1000):
        vocab[f"[unused{i}]"] = i
    vocab["[PAD]"], vocab["[UNK]"], vocab["[CLS]"], vocab["[SEP]"], vocab["[MASK]"] = 0, 100, 101, 102, 103
    vocab = {k: v for k, v in vocab.items() if v not in (0, 100, 101, 102, 103) or not k.startswith("[unused")}
    by_id = {v: k for k, v in vocab.items()}
    words = []
    i = 0
    while len(by_id) < n:
        w = "".join(g.choice(letters, size=int(g.integers(2, 8))))
        tok = w if (len(by_id) % 3) else "##" + w
        if tok in vocab:
            continue
        while i in by_id:
            i += 1
        vocab[tok] = i
        by_id[i] = tok
        if not tok.startswith("##"):
            words.append(w)
    return vocab, words


def bench_from_text(enc: DeviceEncoder, shard: torch.Tensor, n_chunks: int = 1024, batch: int = 256):
    """Extra: chunks/s from RAW TEXT -- native multi-threaded WordPiece on the host, one batch tokenised ahead
    while the GPU encodes the previous one, H2D of the packed ids, encoder forward, append to the shard."""
    import threading

    from multimodal_rag_amd.tokenizer import NativeWordPieceTokenizer

    vocab, words = synthetic_vocab(enc.cfg.vocab)
    tk = NativeWordPieceTokenizer(vocab)
    g = np.random.default_rng(5)
    texts = []
    for _ in range(n_chunks):
        ws = g.choice(len(words), size=260)
        t = " ".join(words[j] + ("ing" if k % 7 == 0 else "") + ("." if k % 13 == 12 else "") for k, j in enumerate(ws))
        texts.append(t[:1000].capitalize())
    batches = [texts[i:i + batch] for i in range(0, n_chunks, batch)]
    L = enc.cfg.max_seq_length if enc.cfg.max_seq_length <= SEQ else SEQ

    def go():
        nxt = {}

        def tok(i):
            nxt[i] = tk.encode_batch_arrays(batches[i], L)

        tok(0)
        n_tok = 0
        for i in range(len(batches)):
            th = None
            if i + 1 < len(batches):
                th = threading.Thread(target=tok, args=(i + 1,))
                th.start()
            ids, lens = nxt.pop(i)
            n_tok += int(lens.sum())
            out = enc.encode_id_rows(ids, lens)
            _native.append_rows(shard, i * batch, out, enc.cfg.dim)
            if th is not None:
                th.join()
        torch.cuda.synchronize()
        return n_tok

    go()
    t0 = time.perf_counter()
    n_tok = go()
    dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    for b in batches:
        tk.encode_batch_arrays(b, L)
    dt_tok = time.perf_counter() - t0
    return {"chunks_per_s": round(n_chunks / dt, 1), "mean_tokens_per_chunk": round(n_tok / n_chunks, 1),
            "tokenizer_alone_chunks_per_s": round(n_chunks / dt_tok, 1), "tokenizer_threads": tk.n_threads,
            "note": "raw ~1000-char texts -> native WordPiece (synthetic vocabulary) -> H2D -> forward -> append; "
                    "tokenisation of batch i+1 overlaps the GPU work of batch i"}


def bench_clip_images(dev, n_images: int = 256, steps: int = 5):
    """Extra (BASELINE config 4): CLIP ViT-B/32 vision tower, uint8 224x224 tiles -> 512-d, images/s."""
    from multimodal_rag_amd.clip import VIT_B32, DeviceClip

    clip = DeviceClip.random_init(VIT_B32, seed=7, device=dev)
    g = torch.Generator(device=dev).manual_seed(3)
    tiles = torch.randint(0, 256, (n_images, 224, 224, 3), generator=g, device=dev, dtype=torch.uint8)
    out = torch.empty((n_images, VIT_B32.proj), dtype=torch.float32, device=dev)
    for _ in range(2):
        clip.encode_images(tiles, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        clip.encode_images(tiles, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    tf = n_images * clip.image_flops() / (ms * 1e-3) / 1e12
    return {"images_per_s": round(n_images / ms * 1e3, 1), "ms_per_step": round(ms, 3), "images_per_step": n_images,
            "tflops": round(tf, 1), "note": "vision tower incl. fused uint8 preprocessing, random-init fp16"}
