"""Golden ids for ClipBpeTokenizer from transformers.CLIPTokenizer on a SYNTHETIC vocabulary.

The trained CLIP vocabulary cannot be fetched here, so a small byte-level BPE (same file formats as
openai/clip-vit-base-patch32: vocab.json, merges.txt) is trained on data/sample text by the plain BPE
procedure and written next to this script; transformers' tokenizer then produces the expected ids.
Run:  python tests/golden/make_clip_bpe_golden.py
"""
import collections
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from multimodal_rag_amd.tokenizer import _bytes_to_unicode  # noqa: E402

import regex  # noqa: E402

PAT = regex.compile(r"'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+")

TEXTS = [
    "A photo of a cat",
    "a diagram of the retrieval pipeline, with 3 stages!",
    "Retrieval-Augmented Generation (RAG) combines search and language models.",
    "it's the user's question: what's new in v2.0?",
    "   multiple   spaces\tand\nnewlines   ",
    "naïve café — déjà vu … ünïcödé ✓",
    "数据检索 и поиск 123456",
    "<|startoftext|>explicit specials<|endoftext|> trailing",
    "",
    "word " * 120,
]


def train(corpus: str, n_merges: int):
    b2u = _bytes_to_unicode()
    words = collections.Counter()
    for tok in PAT.findall(" ".join(corpus.split()).lower()):
        sym = [b2u[b] for b in tok.encode("utf-8")]
        sym[-1] += "</w>"
        words[tuple(sym)] += 1
    merges = []
    for _ in range(n_merges):
        pairs = collections.Counter()
        for w, c in words.items():
            for a, b in zip(w, w[1:]):
                pairs[(a, b)] += c
        if not pairs:
            break
        (a, b), c = max(pairs.items(), key=lambda kv: (kv[1], kv[0]))
        if c < 2:
            break
        merges.append((a, b))
        new = collections.Counter()
        for w, cnt in words.items():
            out, i = [], 0
            while i < len(w):
                if i + 1 < len(w) and w[i] == a and w[i + 1] == b:
                    out.append(a + b)
                    i += 2
                else:
                    out.append(w[i])
                    i += 1
            new[tuple(out)] += cnt
        words = new
    return merges


def main():
    sample = open(os.path.join(HERE, "sample_document.txt"), encoding="utf-8").read()
    corpus = sample + " " + " ".join(TEXTS[:5])
    merges = train(corpus, 400)
    b2u = _bytes_to_unicode()
    chars = [b2u[b] for b in range(256)]
    tokens = chars + [c + "</w>" for c in chars] + [a + b for a, b in merges] + ["<|startoftext|>", "<|endoftext|>"]
    vocab = {}
    for t in tokens:
        vocab.setdefault(t, len(vocab))
    merge_lines = [f"{a} {b}" for a, b in merges]
    json.dump(vocab, open(os.path.join(HERE, "clip_bpe_vocab.json"), "w", encoding="utf-8"), ensure_ascii=False)
    with open(os.path.join(HERE, "clip_bpe_merges.txt"), "w", encoding="utf-8") as f:
        f.write("#version: 0.2\n" + "\n".join(merge_lines) + "\n")

    from transformers import CLIPTokenizer

    tk = CLIPTokenizer(vocab=vocab, merges=[tuple(m.split()) for m in merge_lines])
    cases = []
    for t in TEXTS:
        cases.append({"text": t, "ids": tk(t, truncation=True, max_length=77)["input_ids"],
                      "ids_16": tk(t, truncation=True, max_length=16)["input_ids"]})
    json.dump({"generator": "transformers.CLIPTokenizer " + __import__("transformers").__version__,
               "sot": vocab["<|startoftext|>"], "eot": vocab["<|endoftext|>"], "cases": cases},
              open(os.path.join(HERE, "clip_bpe_expected.json"), "w", encoding="utf-8"), ensure_ascii=False, indent=0)
    print(len(vocab), "tokens,", len(merges), "merges,", len(cases), "cases")


if __name__ == "__main__":
    main()
