"""Host-side tokenisation for the encoder (the step before the hot path).

`WordPieceTokenizer` restates BERT's uncased BasicTokenizer + WordPiece (what
sentence-transformers runs for all-MiniLM-L6-v2 / bge) over a user-supplied LOCAL vocab.txt.
`HashTokenizer` is a clearly-labelled STAND-IN for when no vocabulary is available (this
environment has none and cannot fetch one): deterministic, same id range and special tokens,
but not a trained vocabulary.
"""
from __future__ import annotations

import hashlib
import unicodedata
from typing import Dict, List

CLS, SEP, PAD, UNK = 101, 102, 0, 100


def _is_punct(ch: str) -> bool:
    cp = ord(ch)
    if (33 <= cp <= 47) or (58 <= cp <= 64) or (91 <= cp <= 96) or (123 <= cp <= 126):
        return True
    return unicodedata.category(ch).startswith("P")


def _is_cjk(cp: int) -> bool:
    return ((0x4E00 <= cp <= 0x9FFF) or (0x3400 <= cp <= 0x4DBF) or (0x20000 <= cp <= 0x2A6DF)
            or (0x2A700 <= cp <= 0x2B73F) or (0x2B740 <= cp <= 0x2B81F) or (0x2B820 <= cp <= 0x2CEAF)
            or (0xF900 <= cp <= 0xFAFF) or (0x2F800 <= cp <= 0x2FA1F))


def basic_tokenize(text: str, lower: bool = True) -> List[str]:
    """BERT BasicTokenizer: clean, space CJK, whitespace split, lower + strip accents, split punctuation."""
    out = []
    for ch in text:
        cp = ord(ch)
        if cp == 0 or cp == 0xFFFD or (unicodedata.category(ch) in ("Cc", "Cf") and ch not in "\t\n\r"):
            continue
        if _is_cjk(cp):
            out.append(f" {ch} ")
        elif ch in " \t\n\r" or unicodedata.category(ch) == "Zs":
            out.append(" ")
        else:
            out.append(ch)
    words = "".join(out).split()
    tokens: List[str] = []
    for w in words:
        if lower:
            w = w.lower()
            w = "".join(c for c in unicodedata.normalize("NFD", w) if unicodedata.category(c) != "Mn")
        cur = ""
        for ch in w:
            if _is_punct(ch):
                if cur:
                    tokens.append(cur)
                    cur = ""
                tokens.append(ch)
            else:
                cur += ch
        if cur:
            tokens.append(cur)
    return tokens


class WordPieceTokenizer:
    def __init__(self, vocab: Dict[str, int], lower: bool = True, max_chars_per_word: int = 100):
        self.vocab = vocab
        self.lower = lower
        self.max_chars = max_chars_per_word
        self.cls = vocab.get("[CLS]", CLS)
        self.sep = vocab.get("[SEP]", SEP)
        self.unk = vocab.get("[UNK]", UNK)
        self.vocab_size = max(vocab.values()) + 1

    @classmethod
    def from_vocab_file(cls, path: str, lower: bool = True) -> "WordPieceTokenizer":
        with open(path, encoding="utf-8") as f:
            vocab = {line.rstrip("\n"): i for i, line in enumerate(f)}
        return cls(vocab, lower)

    def _wordpiece(self, word: str) -> List[int]:
        if len(word) > self.max_chars:
            return [self.unk]
        ids, start = [], 0
        while start < len(word):
            end, cur = len(word), None
            while start < end:
                piece = word[start:end] if start == 0 else "##" + word[start:end]
                if piece in self.vocab:
                    cur = self.vocab[piece]
                    break
                end -= 1
            if cur is None:
                return [self.unk]
            ids.append(cur)
            start = end
        return ids

    def encode(self, text: str, max_length: int) -> List[int]:
        ids = [self.cls]
        for w in basic_tokenize(text, self.lower):
            ids.extend(self._wordpiece(w))
            if len(ids) >= max_length - 1:
                break
        return ids[: max_length - 1] + [self.sep]


def _utf32(strings):
    """(codepoints uint32 [total], offsets int64 [n+1]) of a list of str (lone surrogates pass through)."""
    import numpy as np

    blobs = [s.encode("utf-32-le", "surrogatepass") for s in strings]
    offsets = np.zeros(len(blobs) + 1, np.int64)
    np.cumsum([len(b) // 4 for b in blobs], out=offsets[1:])
    cps = np.frombuffer(b"".join(blobs), dtype=np.uint32) if blobs else np.zeros(0, np.uint32)
    return np.ascontiguousarray(cps), offsets


class NativeWordPieceTokenizer:
    """The same tokenisation as `WordPieceTokenizer`, done by libmmrag.so's multi-threaded C++ routine
    (csrc/tokenizer.cpp): ingest needs tens of thousands of chunks per second, the Python loop gives ~500 per
    core.  Host-only: works without a GPU.  `encode` keeps the single-text interface; `encode_batch` is the one to
    use in front of the encoder."""

    def __init__(self, vocab: Dict[str, int], lower: bool = True, n_threads: int = 0):
        import os

        import numpy as np

        from . import _native

        self._lib = _native.lib()
        n = max(vocab.values()) + 1
        tokens = [""] * n
        for t, i in vocab.items():           # position = id; a gap stays an empty (never matched) entry
            tokens[i] = t
        if len(vocab) != n:
            raise ValueError("vocabulary ids must be 0..n-1 without gaps")
        cps, offs = _utf32(tokens)
        self._h = self._lib.mmrag_wordpiece_create(cps.ctypes.data, offs.ctypes.data, n, int(lower))
        if not self._h:
            raise RuntimeError(self._lib.mmrag_last_error().decode())
        self.vocab_size = n
        self.cls = vocab.get("[CLS]", CLS)
        self.sep = vocab.get("[SEP]", SEP)
        self.n_threads = n_threads or min(32, os.cpu_count() or 1)
        self._np = np

    @classmethod
    def from_vocab_file(cls, path: str, lower: bool = True, n_threads: int = 0) -> "NativeWordPieceTokenizer":
        with open(path, encoding="utf-8") as f:
            vocab = {line.rstrip("\n"): i for i, line in enumerate(f)}
        return cls(vocab, lower, n_threads)

    def encode_batch_arrays(self, texts: List[str], max_length: int):
        """(ids [n, max_length] int32, lens [n] int32): rows are [CLS] ... [SEP]; entries past lens[i] are
        unspecified.  The form `DeviceEncoder.encode_id_rows` takes."""
        np = self._np
        cps, offs = _utf32(texts)
        ids = np.empty((len(texts), max_length), np.int32)
        lens = np.empty(len(texts), np.int32)
        st = self._lib.mmrag_wordpiece_encode_batch(self._h, cps.ctypes.data, offs.ctypes.data, len(texts), max_length,
                                                    ids.ctypes.data, lens.ctypes.data, self.n_threads)
        if st:
            raise RuntimeError(self._lib.mmrag_last_error().decode())
        return ids, lens

    def encode_batch(self, texts: List[str], max_length: int) -> List[List[int]]:
        np = self._np
        cps, offs = _utf32(texts)
        ids = np.empty((len(texts), max_length), np.int32)
        lens = np.empty(len(texts), np.int32)
        st = self._lib.mmrag_wordpiece_encode_batch(self._h, cps.ctypes.data, offs.ctypes.data, len(texts), max_length,
                                                    ids.ctypes.data, lens.ctypes.data, self.n_threads)
        if st:
            raise RuntimeError(self._lib.mmrag_last_error().decode())
        return [ids[i, : lens[i]].tolist() for i in range(len(texts))]

    def encode(self, text: str, max_length: int) -> List[int]:
        return self.encode_batch([text], max_length)[0]

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        lib = getattr(self, "_lib", None)
        if h and lib is not None:
            try:
                lib.mmrag_wordpiece_destroy(h)
            except Exception:  # interpreter shutdown: the library may already be gone
                pass


class HashTokenizer:
    """STAND-IN tokenizer (no trained vocabulary available offline): BERT basic tokenisation, then
    each word maps to id 1000 + md5(word) mod (vocab - 1000).  [CLS]=101, [SEP]=102."""

    def __init__(self, vocab_size: int = 30522, lower: bool = True):
        self.vocab_size = vocab_size
        self.lower = lower
        self.cls, self.sep = CLS, SEP

    def encode(self, text: str, max_length: int) -> List[int]:
        ids = [self.cls]
        span = self.vocab_size - 1000
        for w in basic_tokenize(text, self.lower)[: max_length - 2]:
            ids.append(1000 + int.from_bytes(hashlib.md5(w.encode("utf-8")).digest()[:8], "little") % span)
        return ids + [self.sep]


# ---------------------------------------------------------------------------------------------
# CLIP text tower: byte-level BPE (BASELINE config 4; no reference behaviour)
# ---------------------------------------------------------------------------------------------
def _bytes_to_unicode() -> Dict[int, str]:
    """GPT-2's reversible byte -> printable character table (published with the CLIP tokenizer)."""
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(0xA1, 0xAC + 1)) + list(range(0xAE, 0xFF + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return {b: chr(c) for b, c in zip(bs, cs)}


class ClipBpeTokenizer:
    """CLIP's lower-cased byte-level BPE over user-supplied LOCAL `vocab.json` + `merges.txt`
    (the files that ship with openai/clip-vit-base-patch32; none can be fetched here).

    Pipeline (as `transformers.CLIPTokenizer`, which the tests compare against on a synthetic
    vocabulary): NFC -> collapse whitespace -> lower -> split with CLIP's pattern -> bytes to the
    printable table -> merge pairs by rank with `</w>` on the last symbol -> ids;
    `[<|startoftext|>] + ids[:max_length-2] + [<|endoftext|>]`.  ftfy's text repair (optional in
    the original) is not applied."""

    PATTERN = r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+"

    def __init__(self, vocab: Dict[str, int], merges: List[str], context_length: int = 77):
        import regex

        self.vocab = vocab
        self.ranks = {}
        for i, line in enumerate(merges):
            parts = line.split()
            if len(parts) == 2:
                self.ranks[(parts[0], parts[1])] = i
        self.sot = vocab["<|startoftext|>"]
        self.eot = vocab["<|endoftext|>"]
        self.context_length = context_length
        self.byte_chars = _bytes_to_unicode()
        self._pat = regex.compile(self.PATTERN)
        self._ws = regex.compile(r"\s+")
        self._cache: Dict[str, List[int]] = {}

    @classmethod
    def from_files(cls, vocab_json: str, merges_txt: str, context_length: int = 77) -> "ClipBpeTokenizer":
        import json

        with open(vocab_json, encoding="utf-8") as f:
            vocab = json.load(f)
        with open(merges_txt, encoding="utf-8") as f:
            lines = [ln.rstrip("\n") for ln in f]
        if lines and lines[0].startswith("#"):
            lines = lines[1:]
        return cls(vocab, [ln for ln in lines if ln.strip()], context_length)

    def _bpe(self, token: str) -> List[int]:
        hit = self._cache.get(token)
        if hit is not None:
            return hit
        word = [self.byte_chars[b] for b in token.encode("utf-8")]
        word[-1] += "</w>"
        while len(word) > 1:
            best_rank, best_i = None, -1
            for i in range(len(word) - 1):
                r = self.ranks.get((word[i], word[i + 1]))
                if r is not None and (best_rank is None or r < best_rank):
                    best_rank, best_i = r, i
            if best_rank is None:
                break
            first, second = word[best_i], word[best_i + 1]
            merged, i = [], 0
            while i < len(word):  # merge every occurrence of the best pair, left to right
                if i + 1 < len(word) and word[i] == first and word[i + 1] == second:
                    merged.append(first + second)
                    i += 2
                else:
                    merged.append(word[i])
                    i += 1
            word = merged
        ids = [self.vocab.get(w, self.eot) for w in word]  # unk_token is <|endoftext|>
        self._cache[token] = ids
        return ids

    def encode(self, text: str, max_length: int = 0) -> List[int]:
        max_length = max_length or self.context_length
        text = self._ws.sub(" ", unicodedata.normalize("NFC", text)).lower()
        ids: List[int] = []
        for tok in self._pat.findall(text):
            if tok == "<|startoftext|>":
                ids.append(self.sot)
            elif tok == "<|endoftext|>":
                ids.append(self.eot)
            else:
                ids.extend(self._bpe(tok))
        return [self.sot] + ids[: max(0, max_length - 2)] + [self.eot]


class NativeClipBpeTokenizer:
    """The same tokenisation as `ClipBpeTokenizer`, done by libmmrag.so's multi-threaded C++ routine
    (csrc/clip_bpe.cpp): NFC, whitespace collapse and lower-casing stay here (C-speed str methods), the pattern
    split, byte mapping, merges and lookup are native.  Host-only: works without a GPU."""

    def __init__(self, vocab: Dict[str, int], merges: List[str], context_length: int = 77, n_threads: int = 0):
        import os

        import numpy as np
        import regex

        from . import _native

        self._lib = _native.lib()
        toks = list(vocab.keys())
        v_cps, v_offs = _utf32(toks)
        v_ids = np.asarray([vocab[t] for t in toks], dtype=np.int32)
        good = [ln for ln in merges if len(ln.split()) == 2]
        m_cps, m_offs = _utf32([" ".join(ln.split()) for ln in good])
        self._h = self._lib.mmrag_clip_bpe_create(v_cps.ctypes.data, v_offs.ctypes.data, v_ids.ctypes.data, len(toks),
                                                  m_cps.ctypes.data, m_offs.ctypes.data, len(good))
        if not self._h:
            raise RuntimeError(self._lib.mmrag_last_error().decode())
        self.sot, self.eot = vocab["<|startoftext|>"], vocab["<|endoftext|>"]
        self.context_length = context_length
        self.n_threads = n_threads or min(32, os.cpu_count() or 1)
        self._ws = regex.compile(r"\s+")
        self._np = np

    @classmethod
    def from_files(cls, vocab_json: str, merges_txt: str, context_length: int = 77, n_threads: int = 0):
        py = ClipBpeTokenizer.from_files(vocab_json, merges_txt, context_length)
        with open(merges_txt, encoding="utf-8") as f:
            lines = [ln.rstrip("\n") for ln in f]
        if lines and lines[0].startswith("#"):
            lines = lines[1:]
        return cls(py.vocab, [ln for ln in lines if ln.strip()], context_length, n_threads)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.mmrag_clip_bpe_destroy(h)

    def encode_batch_arrays(self, texts: List[str], max_length: int = 0):
        """(ids [n, max_length] int32, lens [n] int32): rows are [sot] ... [eot]; entries past lens[i] are unspecified."""
        np = self._np
        max_length = max_length or self.context_length
        cps, offs = _utf32([self._ws.sub(" ", unicodedata.normalize("NFC", t)).lower() for t in texts])
        ids = np.empty((len(texts), max_length), np.int32)
        lens = np.empty(len(texts), np.int32)
        st = self._lib.mmrag_clip_bpe_encode_batch(self._h, cps.ctypes.data, offs.ctypes.data, len(texts), max_length,
                                                   ids.ctypes.data, lens.ctypes.data, self.n_threads)
        if st:
            raise RuntimeError(self._lib.mmrag_last_error().decode())
        return ids, lens

    def encode_batch(self, texts: List[str], max_length: int = 0) -> List[List[int]]:
        ids, lens = self.encode_batch_arrays(texts, max_length)
        return [ids[i, : lens[i]].tolist() for i in range(len(texts))]

    def encode(self, text: str, max_length: int = 0) -> List[int]:
        return self.encode_batch([text], max_length)[0]
