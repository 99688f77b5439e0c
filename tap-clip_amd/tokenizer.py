"""CLIP's byte-pair-encoding tokenizer (the algorithm open_clip's `SimpleTokenizer` implements, written
from its published description), for users who have the vocabulary file
(`bpe_simple_vocab_16e6.txt.gz`, not available offline in this project: `CLIPWrapper` falls back to the
deterministic `HashTokenizer` without it).  `tokenizer(texts) -> [n, 77] int64`, SOT/EOT added, zero
padded, truncated with EOT kept -- what reference models/prompt_learner.py:31-33 consumes.

Parity unpinned (the vocabulary file is absent here, so no id sequence of open_clip's can be reproduced); known
difference: open_clip first runs `ftfy.fix_text` on the input, which is skipped here (ftfy is not installed)."""


def find_bpe_vocab():
    """Path of CLIP's vocabulary file if the user's environment has one: $TAPCLIP_BPE_PATH, or the data file inside
    an installed `open_clip` package (located through importlib, without importing the package).  None otherwise."""
    import importlib.util
    import os

    env = os.environ.get("TAPCLIP_BPE_PATH")
    if env:
        return env
    try:
        spec = importlib.util.find_spec("open_clip")
    except (ImportError, ValueError):
        spec = None
    if spec is not None and spec.submodule_search_locations:
        for d in spec.submodule_search_locations:
            cand = os.path.join(d, "bpe_simple_vocab_16e6.txt.gz")
            if os.path.exists(cand):
                return cand
    return None
import gzip
import html
import re
from functools import lru_cache

try:  # CLIP's pattern uses the Unicode classes \p{L} / \p{N}, which only the third-party `regex` module has
    import regex as _re_u
except ImportError:  # pragma: no cover - ASCII classes: identical on ASCII text, differs on other scripts
    _re_u = None
from typing import Dict, List, Tuple, Union

import torch


@lru_cache()
def _bytes_to_unicode() -> Dict[int, str]:
    """Reversible byte -> printable unicode character table (GPT-2 style)."""
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("\xa1"), ord("\xac") + 1)) + list(range(ord("\xae"), ord("\xff") + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return dict(zip(bs, (chr(c) for c in cs)))


def _clean(text: str) -> str:
    text = html.unescape(html.unescape(text))
    return re.sub(r"\s+", " ", text).strip().lower()


class BPETokenizer:
    def __init__(self, bpe_path: str, context_length: int = 77, n_merges: int = 49152 - 256 - 2):
        opener = gzip.open if bpe_path.endswith(".gz") else open
        with opener(bpe_path, "rt", encoding="utf-8") as fh:
            lines = fh.read().split("\n")
        merges = [tuple(l.split()) for l in lines[1: 1 + n_merges] if len(l.split()) == 2]
        byte_chars = list(_bytes_to_unicode().values())
        vocab = byte_chars + [c + "</w>" for c in byte_chars] + ["".join(m) for m in merges] + ["<start_of_text>", "<end_of_text>"]
        self.encoder = {tok: i for i, tok in enumerate(vocab)}
        self.ranks = {m: i for i, m in enumerate(merges)}
        self.byte_encoder = _bytes_to_unicode()
        self.context_length = context_length
        self.sot, self.eot = self.encoder["<start_of_text>"], self.encoder["<end_of_text>"]
        self.cache: Dict[str, str] = {}
        if _re_u is not None:
            self.pat = _re_u.compile(r"<start_of_text>|<end_of_text>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+",
                                     _re_u.IGNORECASE)
        else:
            self.pat = re.compile(r"<start_of_text>|<end_of_text>|'s|'t|'re|'ve|'m|'ll|'d|[a-zA-Z]+|[0-9]|[^\sa-zA-Z0-9]+")

    def _bpe(self, token: str) -> str:
        if token in self.cache:
            return self.cache[token]
        word: Tuple[str, ...] = tuple(token[:-1]) + (token[-1] + "</w>",)
        while len(word) > 1:
            pairs = {(word[i], word[i + 1]) for i in range(len(word) - 1)}
            best = min(pairs, key=lambda p: self.ranks.get(p, float("inf")))
            if best not in self.ranks:
                break
            a, b = best
            out: List[str] = []
            i = 0
            while i < len(word):
                if i < len(word) - 1 and word[i] == a and word[i + 1] == b:
                    out.append(a + b)
                    i += 2
                else:
                    out.append(word[i])
                    i += 1
            word = tuple(out)
        res = " ".join(word)
        self.cache[token] = res
        return res

    def encode(self, text: str) -> List[int]:
        ids: List[int] = []
        for tok in self.pat.findall(_clean(text)):
            tok = "".join(self.byte_encoder[b] for b in tok.encode("utf-8"))
            ids.extend(self.encoder[t] for t in self._bpe(tok).split(" "))
        return ids

    def __call__(self, texts: Union[str, List[str]], context_length: int = None) -> torch.Tensor:
        if isinstance(texts, str):
            texts = [texts]
        L = context_length or self.context_length
        out = torch.zeros(len(texts), L, dtype=torch.long)
        for i, t in enumerate(texts):
            ids = [self.sot] + self.encode(t) + [self.eot]
            if len(ids) > L:
                ids = ids[:L]
                ids[-1] = self.eot
            out[i, : len(ids)] = torch.tensor(ids)
        return out
