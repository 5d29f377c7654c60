"""Byte-level BPE tokenizer of the ``open_clip`` flavour -- behaviour of ``SimpleTokenizer`` / ``openclip_tokenize`` in
easynlp/modelzoo/models/clip/openclip_tokenizer.py:24-150 and easynlp/appzoo/clip/data.py:137-161 (the CLIP tokenizer of
Radford et al. 2021: GPT-2's reversible byte <-> unicode table, an end-of-word marker ``</w>``, ranked merges).

The checkpoint's ``vocab.txt`` is the gzip-compressed merges file (openclip_tokenizer.py:75): line 0 is a header, lines
1 .. 48894 are ``left right`` merges in rank order.  Vocabulary ids: the 256 byte symbols, the same with ``</w>``, one entry
per merge, then ``<start_of_text>`` and ``<end_of_text>`` -- so the EOT id is the largest, which is what the text tower's
arg-max pooling relies on (modeling_openclip.py:366).

Text cleaning: ``ftfy.fix_text`` when ftfy is installed (the reference imports it unconditionally; it is not part of this
image, where the step is skipped), ``html.unescape`` twice, whitespace collapsed, lower-cased.  Pinned to the reference
implementation by tests/test_bpe_tokenizer.py (fixture + live fuzz comparison).
"""
from __future__ import annotations

import gzip
import html
from typing import Dict, List, Tuple, Union

import regex
import torch

MAX_MERGES = 49152 - 256 - 2          # openclip_tokenizer.py:76
SOT, EOT = "<start_of_text>", "<end_of_text>"
END = "</w>"


def byte_symbols() -> List[str]:
    """symbol of each byte value 0..255: printable latin-1 bytes stand for themselves, the other 68 get code points from 256
    upwards in byte order (GPT-2's reversible table, so no symbol is whitespace or a control character)"""
    keep = set(range(ord("!"), ord("~") + 1)) | set(range(0xA1, 0xAC + 1)) | set(range(0xAE, 0xFF + 1))
    out, extra = [], 0
    for b in range(256):
        if b in keep:
            out.append(chr(b))
        else:
            out.append(chr(256 + extra))
            extra += 1
    return out


def _vocab_order(symbols: List[str]) -> List[str]:
    """the reference enumerates the table as dict values of a list that starts with the kept bytes and appends the
    remapped ones: ids follow THAT order (kept bytes ascending, then the remapped bytes ascending)"""
    kept = [s for b, s in enumerate(symbols) if ord(s) == b]
    moved = [s for b, s in enumerate(symbols) if ord(s) != b]
    return kept + moved


def _clean(text: str) -> str:
    try:
        import ftfy
        text = ftfy.fix_text(text)
    except ImportError:
        pass
    text = html.unescape(html.unescape(text)).strip()
    return regex.sub(r"\s+", " ", text).strip().lower()


class SimpleTokenizer:

    def __init__(self, bpe_path: str, special_tokens=None):
        self.symbols = byte_symbols()
        with gzip.open(bpe_path) as f:
            lines = f.read().decode("utf-8").split("\n")
        merges: List[Tuple[str, ...]] = [tuple(line.split()) for line in lines[1:MAX_MERGES + 1]]
        base = _vocab_order(self.symbols)
        vocab = base + [s + END for s in base] + ["".join(m) for m in merges]
        specials = [SOT, EOT] + list(special_tokens or [])
        vocab += specials
        self.encoder: Dict[str, int] = dict(zip(vocab, range(len(vocab))))
        self.decoder = {i: t for t, i in self.encoder.items()}
        self.rank: Dict[Tuple[str, ...], int] = dict(zip(merges, range(len(merges))))
        self.special_tokens = specials
        self.vocab_size = len(self.encoder)
        self.all_special_ids = [self.encoder[t] for t in specials]
        self._cache: Dict[str, List[str]] = {t: [t] for t in specials}
        self._pattern = regex.compile("|".join(specials) + r"""|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+""",
                                      regex.IGNORECASE)

    def _merge_word(self, token: str) -> List[str]:
        """ranked merges over the symbols of one pre-token (last symbol carries the end-of-word marker)"""
        hit = self._cache.get(token)
        if hit is not None:
            return hit
        parts = list(token[:-1]) + [token[-1] + END]
        while len(parts) > 1:
            best, best_rank = None, None
            for pair in zip(parts[:-1], parts[1:]):
                r = self.rank.get(pair)
                if r is not None and (best_rank is None or r < best_rank):
                    best, best_rank = pair, r
            if best is None:
                break
            merged, i = [], 0
            while i < len(parts):
                if i + 1 < len(parts) and parts[i] == best[0] and parts[i + 1] == best[1]:
                    merged.append(parts[i] + parts[i + 1])
                    i += 2
                else:
                    merged.append(parts[i])
                    i += 1
            parts = merged
        self._cache[token] = parts
        return parts

    def encode(self, text: str) -> List[int]:
        ids: List[int] = []
        for tok in self._pattern.findall(_clean(text)):
            mapped = "".join(self.symbols[b] for b in tok.encode("utf-8"))
            ids.extend(self.encoder[p] for p in self._merge_word(mapped))
        return ids

    def decode(self, tokens) -> str:
        inverse = {s: b for b, s in enumerate(self.symbols)}
        text = "".join(self.decoder[int(t)] for t in tokens)
        return bytearray(inverse[c] for c in text).decode("utf-8", errors="replace").replace(END, " ")


def openclip_tokenize(texts: Union[str, List[str]], context_length: int = 77, _tokenizer: SimpleTokenizer = None) -> torch.Tensor:
    """``[SOT] + bpe(text) + [EOT]`` per text, cut to ``context_length`` (a longer caption loses its EOT, exactly as
    appzoo/clip/data.py:153-159 does), zero padded; int64 [len(texts), context_length]."""
    if isinstance(texts, str):
        texts = [texts]
    sot, eot = _tokenizer.encoder[SOT], _tokenizer.encoder[EOT]
    out = torch.zeros(len(texts), context_length, dtype=torch.long)
    for i, t in enumerate(texts):
        ids = ([sot] + _tokenizer.encode(t) + [eot])[:context_length]
        out[i, :len(ids)] = torch.tensor(ids)
    return out
