"""CPU: WordPiece restatement vs transformers.BertTokenizer on a locally built vocabulary."""
import pytest

from multimodal_rag_amd.tokenizer import HashTokenizer, WordPieceTokenizer, basic_tokenize

VOCAB = ["[PAD]"] + [f"[unused{i}]" for i in range(99)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"] + [
    "machine", "learning", "la", "gi", "?", "!", ",", ".", "un", "##aff", "##able", "##ing", "run", "##s",
    "the", "quick", "brown", "fox", "cafe", "naive", "你", "好", "-", "state", "of", "art", "##ly", "embed", "##ding"]
TEXTS = ["Machine learning là gì?", "The quick brown fox runs.", "unaffable running, naïve café!",
         "state-of-the-art embeddingly 你好", "", "   ", "xyzzy " * 3, "a" * 120]


def test_basic_tokenize_examples():
    assert basic_tokenize("Hello, World!") == ["hello", ",", "world", "!"]
    assert basic_tokenize("naïve café") == ["naive", "cafe"]
    assert basic_tokenize("你好") == ["你", "好"]


def test_wordpiece_matches_transformers(tmp_path):
    transformers = pytest.importorskip("transformers")
    p = tmp_path / "vocab.txt"
    p.write_text("\n".join(VOCAB) + "\n", encoding="utf-8")
    ref = transformers.BertTokenizer(str(p), do_lower_case=True)
    mine = WordPieceTokenizer.from_vocab_file(str(p))
    for t in TEXTS:
        want = ref(t, truncation=True, max_length=32)["input_ids"]
        assert mine.encode(t, 32) == want, t
    long = "the quick brown fox " * 50
    assert mine.encode(long, 16) == ref(long, truncation=True, max_length=16)["input_ids"]


def test_hash_tokenizer_is_deterministic_and_in_range():
    t = HashTokenizer(30522)
    a = t.encode("Machine learning là gì?", 256)
    assert a == t.encode("machine LEARNING la gi ?", 256)
    assert a[0] == 101 and a[-1] == 102 and all(1000 <= x < 30522 for x in a[1:-1])
    assert len(t.encode("word " * 1000, 256)) == 256
    assert t.encode("", 256) == [101, 102]


# ---- CLIP byte-level BPE -------------------------------------------------------------------
import json
import os

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _clip_tok():
    from multimodal_rag_amd.tokenizer import ClipBpeTokenizer

    return ClipBpeTokenizer.from_files(os.path.join(GOLD, "clip_bpe_vocab.json"), os.path.join(GOLD, "clip_bpe_merges.txt"))


def test_clip_bpe_matches_committed_golden():
    """ids produced by transformers.CLIPTokenizer on the synthetic vocabulary (make_clip_bpe_golden.py)"""
    exp = json.load(open(os.path.join(GOLD, "clip_bpe_expected.json"), encoding="utf-8"))
    tk = _clip_tok()
    assert (tk.sot, tk.eot) == (exp["sot"], exp["eot"])
    for c in exp["cases"]:
        assert tk.encode(c["text"]) == c["ids"], c["text"]
        assert tk.encode(c["text"], 16) == c["ids_16"], c["text"]


def test_clip_bpe_matches_transformers_live():
    transformers = pytest.importorskip("transformers")
    tk = _clip_tok()
    vocab = json.load(open(os.path.join(GOLD, "clip_bpe_vocab.json"), encoding="utf-8"))
    merges = [tuple(ln.split()) for ln in open(os.path.join(GOLD, "clip_bpe_merges.txt"), encoding="utf-8")
              if ln.strip() and not ln.startswith("#")]
    ref = transformers.CLIPTokenizer(vocab=vocab, merges=merges)
    for t in ["Figure 3: throughput vs. batch size", "don't you'll we've I'm", "tab\tsep  x1y2z3", "ÀÉÎ õ ß", "🙂 emoji"]:
        assert tk.encode(t) == ref(t, truncation=True, max_length=77)["input_ids"], t


def test_clip_bpe_shape_rules():
    tk = _clip_tok()
    ids = tk.encode("retrieval " * 200)
    assert len(ids) == 77 and ids[0] == tk.sot and ids[-1] == tk.eot
    assert tk.encode("") == [tk.sot, tk.eot]
    assert tk.eot == max(tk.vocab.values())  # EOS pooling picks argmax(id): the end token must be the largest id


# ---- native (C++) CLIP BPE (csrc/clip_bpe.cpp): must equal the Python form above, which is pinned to transformers -----
def _native_clip_tok(n_threads=3):
    from multimodal_rag_amd.tokenizer import NativeClipBpeTokenizer

    return NativeClipBpeTokenizer.from_files(os.path.join(GOLD, "clip_bpe_vocab.json"),
                                             os.path.join(GOLD, "clip_bpe_merges.txt"), n_threads=n_threads)


def test_native_clip_bpe_matches_committed_golden():
    exp = json.load(open(os.path.join(GOLD, "clip_bpe_expected.json"), encoding="utf-8"))
    nat = _native_clip_tok()
    assert (nat.sot, nat.eot) == (exp["sot"], exp["eot"])
    texts = [c["text"] for c in exp["cases"]]
    assert nat.encode_batch(texts) == [c["ids"] for c in exp["cases"]]
    assert nat.encode_batch(texts, 16) == [c["ids_16"] for c in exp["cases"]]
    ids, lens = nat.encode_batch_arrays(["retrieval " * 200, ""])
    assert ids.shape == (2, 77) and lens.tolist() == [77, 2] and ids[0, 0] == nat.sot and ids[0, 76] == nat.eot


def test_native_clip_bpe_fuzz_against_python():
    """pattern corner cases: contractions next to punctuation, the special tokens inside text, digits one by one,
    letters / numbers / marks of many scripts, every whitespace class, emoji and astral planes, NFC recomposition"""
    import random

    py, nat = _clip_tok(), _native_clip_tok()
    pieces = ["don't", "you'll", "we've", "i'm", "he'd", "it's", "'re", "'", "''s", "!'s", "x'sx", "<|startoftext|>",
              "<|endoftext|>", "<|endoftext", "<|", "|>", "a1b22c333", "12345", "٣٤", "Ⅻ", "½",
              "naïve", "naïve", "Å", "가", "straße", "İstanbul", "ΑΣ",
              "日本語", "テスト", "한국어", "क्षि", "\U0001f600",
              "\U0001f642\U0001f642", "\U0001d518\U0001d52b\U0001d526", " ", " ", "　", "\t", "\n", "\r\n", "\x0b",
              "\x1c", "​", "﻿", "--", "...", "(a)", "[b]", "{c}", "#$%", "retrieval", "throughput", "figure",
              "Figure 3:", "batch-size", "GPU", "mi355x", "the", "quick", "brown"]
    rng = random.Random(5)
    texts = []
    for _ in range(3000):
        k = rng.randint(0, 9)
        parts = [rng.choice(pieces) for _ in range(k)]
        texts.append(rng.choice(["", " ", "  "]).join(parts) if rng.random() < 0.5 else " ".join(parts))
    texts += ["".join(chr(rng.choice([rng.randint(32, 126), rng.randint(0xA0, 0x24F), rng.randint(0x370, 0x3FF),
                                       rng.randint(0x4E00, 0x4E80), rng.randint(0x1F600, 0x1F64F), 32]))
                      for _ in range(rng.randint(0, 60))) for _ in range(1500)]
    for max_len in (77, 9):
        want = [py.encode(t, max_len) for t in texts]
        got = nat.encode_batch(texts, max_len)
        bad = [i for i, (a, b) in enumerate(zip(got, want)) if a != b]
        assert not bad, (texts[bad[0]], got[bad[0]], want[bad[0]])


def test_clip_tables_are_current():
    """csrc/clip_bpe_tables.inc was generated from the regex module installed here"""
    import regex

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    head = open(os.path.join(root, "multimodal_rag_amd", "csrc", "clip_bpe_tables.inc")).readline()
    assert f"regex {regex.__version__}" in head, "re-run tools/gen_clip_tables.py"


# ---- native (C++) WordPiece: must equal the Python restatement, which is pinned to transformers above ---------
def _native_pair(extra=()):
    from multimodal_rag_amd.tokenizer import NativeWordPieceTokenizer

    vocab = {w: i for i, w in enumerate(VOCAB + [e for e in extra if e not in VOCAB])}
    return WordPieceTokenizer(vocab), NativeWordPieceTokenizer(vocab, n_threads=3)


def test_native_wordpiece_matches_transformers(tmp_path):
    transformers = pytest.importorskip("transformers")
    from multimodal_rag_amd.tokenizer import NativeWordPieceTokenizer

    p = tmp_path / "vocab.txt"
    p.write_text("\n".join(VOCAB) + "\n", encoding="utf-8")
    ref = transformers.BertTokenizer(str(p), do_lower_case=True)
    nat = NativeWordPieceTokenizer.from_vocab_file(str(p))
    assert nat.encode_batch(TEXTS, 32) == [ref(t, truncation=True, max_length=32)["input_ids"] for t in TEXTS]
    long = "the quick brown fox " * 50
    assert nat.encode(long, 16) == ref(long, truncation=True, max_length=16)["input_ids"]


def test_native_wordpiece_fuzz_against_python():
    """Unicode corner cases: final sigma, dotted capital I, Hangul and compatibility-ideograph decomposition, stacked
    combining marks (canonical reordering), every whitespace / control class, lone surrogates, astral planes."""
    import random

    extra = ["σ", "ς", "##σ", "ας", "i̇", "ᄀ", "ᅡ", "##ᅡ", "e", "##e", "́", "a", "##a", "b", "##b", "ß", "ss",
             "##s", "豈", "##1", "1"]
    py, nat = _native_pair(extra)
    alphabet = list("abcdeABCDE sS.,!?-'\"#_1 \t\n") + [
        "Σ", "σ", "İ", "I", "é", "É", "ñ", "ü", "Å", "ǅ", "ß", "你", "好", "豈", "가", "한", "́", "̣", "̀", "­", "​",
        " ", " ", "　", " ", " ", "\x00", "�", "\x1c", "\x85", "—", "…", "¿", "«", "༾",
        "\U0001d165", "\U0001d16d", "\U0001f642", "\ud800", "ͅ", "ᾳ", "ΐ", "ﬁ", "Ǆ", "ẞ", "K", "Σ́", "Á̧"]
    rng = random.Random(5)
    cases = list(TEXTS) + ["ΑΣ ΑΣΑ Σ ΑΣ. ΑΣ'Α 'Σ Α.Σ", "İstanbul ISTANBUL", "가나다 한글", "x" * 101 + " ok", "a" * 100]
    cases += ["".join(rng.choice(alphabet) for _ in range(rng.randint(0, 60))) for _ in range(1500)]
    pools = [(0, 0x2FF), (0x300, 0x36F), (0x370, 0x2FFF), (0x3000, 0xFFFF), (0x10000, 0x2FFFF)]
    cases += ["".join(chr(rng.randint(*rng.choice(pools))) for _ in range(rng.randint(1, 30))) for _ in range(800)]
    got = nat.encode_batch(cases, 32)
    for c, g in zip(cases, got):
        assert g == py.encode(c, 32), repr(c)
    for ml in (2, 3, 5):
        for c in cases[:200]:
            assert nat.encode(c, ml) == py.encode(c, ml), (c, ml)
    ids, lens = nat.encode_batch_arrays(cases[:50], 32)
    assert [ids[i, : lens[i]].tolist() for i in range(50)] == got[:50]


def test_unicode_tables_are_current():
    """csrc/unicode_tables.inc was generated from this interpreter's unicodedata"""
    import unicodedata

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    head = open(os.path.join(root, "multimodal_rag_amd", "csrc", "unicode_tables.inc")).readline()
    assert f"unicodedata {unicodedata.unidata_version}" in head, head
