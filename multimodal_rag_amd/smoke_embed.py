"""smoke(): one tiny encoder forward on the GPU checked against the CPU oracle."""
import dataclasses

import numpy as np


def run(dev):
    from oracle import encoder_oracle as E

    from .encoder import DeviceEncoder, EncoderConfig

    shape = E.TINY
    w = E.make_bert_weights(shape, 3)
    cfg = EncoderConfig("tiny", shape.n_layers, shape.hidden, shape.n_heads, shape.intermediate, shape.vocab,
                        shape.max_pos, max_seq_length=shape.max_pos, pool="mean", ln_eps=shape.ln_eps)
    enc = DeviceEncoder(cfg, w, dev)
    g = np.random.default_rng(0)
    seqs = [g.integers(1, shape.vocab, n).tolist() for n in (3, 40, 64)]
    got = enc.encode_ids(seqs).cpu().numpy()
    ref = E.bert_encode(dataclasses.replace(shape, pool="mean"), E.round_weights_fp16(w), seqs)
    err = float(np.abs(got - ref).max())
    assert err <= 4e-3, err
    print("[smoke] embed ok: max |dembedding| = %.2e" % err)
