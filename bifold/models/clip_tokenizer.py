"""CLIP's byte-level BPE tokenizer: `tokenize(texts) -> int32 [N, 77]`, the producer of
`sample["instruction"]` for `rgb_clip` (reference bifold/models/clip/__init__.py:241-287 over
bifold/models/clip/simple_tokenizer.py; pinned by tests/golden/tokenizer.npz, which holds the reference's own
ids for instructions built from its templates).

Host-side string processing, outside the GPU hot path; it exists so that a user of this build can feed text.
The merge table is OpenAI's published `bpe_simple_vocab_16e6.txt.gz` -- a data file the reference vendors and
this repository does not: pass its path, or set BIFOLD_CLIP_BPE to it, or BIFOLD_REFERENCE to a checkout of the
reference.  Unicode repair (`ftfy.fix_text` in the reference) is applied when ftfy is installed and
skipped otherwise (identity on well-formed text).
"""
from __future__ import annotations

import gzip
import html
import os
from typing import Dict, List, Optional, Sequence, Tuple, Union

import torch

try:
    import regex as re
except ImportError as e:            # the token pattern needs \\p{L} / \\p{N}
    raise ImportError("bifold.models.clip_tokenizer needs the `regex` package") from e

N_MERGES = 49152 - 256 - 2
_PATTERN = re.compile(r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+",
                      re.IGNORECASE)


def find_bpe_file(path: Optional[str] = None) -> str:
    cands = [path, os.environ.get("BIFOLD_CLIP_BPE"),
             os.path.join(os.environ.get("BIFOLD_REFERENCE", ""), "bifold", "models", "clip", "bpe_simple_vocab_16e6.txt.gz")]
    for c in cands:
        if c and os.path.isfile(c):
            return c
    raise FileNotFoundError("CLIP BPE merge table not found: pass bpe_path or set BIFOLD_CLIP_BPE to "
                            "bpe_simple_vocab_16e6.txt.gz (OpenAI CLIP; vendored by the reference under bifold/models/clip/)")


def _byte_alphabet() -> Dict[int, str]:
    """GPT-2's reversible byte -> printable-character table: printable Latin-1 bytes map to themselves, the other
    68 byte values to code points 256, 257, ... in increasing byte order."""
    keep = set(range(0x21, 0x7F)) | set(range(0xA1, 0xAD)) | set(range(0xAE, 0x100))
    table, extra = {}, 0
    for b in sorted(keep):
        table[b] = chr(b)
    for b in range(256):
        if b not in keep:
            table[b] = chr(256 + extra)
            extra += 1
    return table


class ClipTokenizer:
    def __init__(self, bpe_path: Optional[str] = None):
        alphabet = _byte_alphabet()
        self.byte_char = alphabet
        lines = gzip.open(find_bpe_file(bpe_path)).read().decode("utf-8").split("\n")
        merges: List[Tuple[str, str]] = [tuple(ln.split()) for ln in lines[1:1 + N_MERGES]]
        # vocabulary order: the 256 byte symbols in (printable-first) alphabet order, the same with the
        # end-of-word mark, one entry per merge, then the two specials
        base = [alphabet[b] for b in list(sorted(b for b in alphabet if alphabet[b] == chr(b))) +
                [b for b in range(256) if alphabet[b] != chr(b)]]
        vocab = base + [s + "</w>" for s in base] + [a + b for a, b in merges] + ["<|startoftext|>", "<|endoftext|>"]
        self.ids = {tok: i for i, tok in enumerate(vocab)}
        self.rank = {pair: i for i, pair in enumerate(merges)}
        self.sot, self.eot = self.ids["<|startoftext|>"], self.ids["<|endoftext|>"]
        self._memo: Dict[str, List[str]] = {}

    def _merge_word(self, token: str) -> List[str]:
        """Greedy BPE: repeatedly join the adjacent pair with the lowest merge rank (every occurrence, left to
        right) until no adjacent pair is in the table."""
        hit = self._memo.get(token)
        if hit is not None:
            return hit
        parts = list(token[:-1]) + [token[-1] + "</w>"]
        while len(parts) > 1:
            best, best_rank = None, None
            for pair in zip(parts, parts[1:]):
                r = self.rank.get(pair)
                if r is not None and (best_rank is None or r < best_rank):
                    best, best_rank = pair, r
            if best is None:
                break
            joined, i = [], 0
            while i < len(parts):
                if i + 1 < len(parts) and parts[i] == best[0] and parts[i + 1] == best[1]:
                    joined.append(parts[i] + parts[i + 1])
                    i += 2
                else:
                    joined.append(parts[i])
                    i += 1
            parts = joined
        self._memo[token] = parts
        return parts

    @staticmethod
    def clean(text: str) -> str:
        try:
            import ftfy
            text = ftfy.fix_text(text)
        except ImportError:
            pass
        text = html.unescape(html.unescape(text)).strip()
        return re.sub(r"\s+", " ", text).strip().lower()

    def encode(self, text: str) -> List[int]:
        out: List[int] = []
        for piece in _PATTERN.findall(self.clean(text)):
            symbols = "".join(self.byte_char[b] for b in piece.encode("utf-8"))
            out.extend(self.ids[p] for p in self._merge_word(symbols))
        return out


_DEFAULT: Optional[ClipTokenizer] = None


def tokenize(texts: Union[str, Sequence[str]], context_length: int = 77, truncate: bool = False,
             bpe_path: Optional[str] = None) -> torch.Tensor:
    """[N, context_length] int32: <start> ids <end>, zero padded; too long raises unless `truncate`."""
    global _DEFAULT
    if _DEFAULT is None or bpe_path is not None:
        tok = ClipTokenizer(bpe_path)
        if bpe_path is None:
            _DEFAULT = tok
    else:
        tok = _DEFAULT
    if isinstance(texts, str):
        texts = [texts]
    result = torch.zeros(len(texts), context_length, dtype=torch.int32)
    for i, text in enumerate(texts):
        ids = [tok.sot] + tok.encode(text) + [tok.eot]
        if len(ids) > context_length:
            if not truncate:
                raise RuntimeError(f"Input {text} is too long for context length {context_length}")
            ids = ids[:context_length]
            ids[-1] = tok.eot
        result[i, :len(ids)] = torch.tensor(ids, dtype=torch.int32)
    return result
