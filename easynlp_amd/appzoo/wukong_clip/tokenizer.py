"""WordPiece tokenizer of the Wukong applications -- behaviour of ``FullTokenizer`` in
easynlp/appzoo/wukong_clip/bert_tokenizer.py (the original BERT ``tokenization.py`` algorithm, Devlin et al. 2018):

1. basic tokenisation: drop NUL / U+FFFD / control characters, map every whitespace character to a blank; put blanks around
   CJK ideographs; split on whitespace; lower-case and strip combining marks (NFD, category Mn); split every punctuation
   character into its own token (ASCII symbols count as punctuation);
2. WordPiece: greedy longest-match-first against the vocabulary, continuation pieces prefixed ``##``; a word longer than
   200 characters, or one that cannot be covered, becomes ``[UNK]``.

Unlike ``transformers.BertTokenizer`` it knows no special tokens inside the text (``"[CLS]"`` in a caption is split like
any other punctuation) and allows 200 characters per word (transformers: 100) -- which is why the Wukong mirrors carry
their own implementation instead of re-using the one the clip dataset uses.  ``tokenize_batch`` is the ``tokenize`` helper
of WukongCLIPDataset / WukongCLIPPredictor (wukong_clip/data.py:181-203): ``[CLS] + ids[:context_length - 2] + [SEP]``,
zero padded.  Pinned to the reference implementation by tests/test_wukong_data.py (fixture + live fuzz comparison).
"""
from __future__ import annotations

import io
import unicodedata
from typing import Dict, Iterable, List, Union

import torch

MAX_INPUT_CHARS_PER_WORD = 200


def _is_whitespace(ch: str) -> bool:
    return ch in (" ", "\t", "\n", "\r") or unicodedata.category(ch) == "Zs"


def _is_control(ch: str) -> bool:
    if ch in ("\t", "\n", "\r"):
        return False
    return unicodedata.category(ch) in ("Cc", "Cf")


def _is_punctuation(ch: str) -> bool:
    cp = ord(ch)
    if 33 <= cp <= 47 or 58 <= cp <= 64 or 91 <= cp <= 96 or 123 <= cp <= 126:
        return True
    return unicodedata.category(ch).startswith("P")


def _is_cjk(cp: int) -> bool:
    return (0x4E00 <= cp <= 0x9FFF or 0x3400 <= cp <= 0x4DBF or 0x20000 <= cp <= 0x2A6DF or 0x2A700 <= cp <= 0x2B73F
            or 0x2B740 <= cp <= 0x2B81F or 0x2B820 <= cp <= 0x2CEAF or 0xF900 <= cp <= 0xFAFF or 0x2F800 <= cp <= 0x2FA1F)


class FullTokenizer:

    def __init__(self, vocab_file: str, do_lower_case: bool = True):
        self.vocab = _read_vocab(vocab_file)
        self.inv_vocab = {i: t for t, i in self.vocab.items()}
        self.do_lower_case = do_lower_case
        self.unk_token = "[UNK]"

    # -- stage 1 ---------------------------------------------------------------------------
    def _basic(self, text: str) -> List[str]:
        cleaned = []
        for ch in text:
            cp = ord(ch)
            if cp == 0 or cp == 0xFFFD or _is_control(ch):
                continue
            if _is_whitespace(ch):
                cleaned.append(" ")
            elif _is_cjk(cp):
                cleaned.extend((" ", ch, " "))
            else:
                cleaned.append(ch)
        out: List[str] = []
        for word in "".join(cleaned).split():
            if self.do_lower_case:
                word = word.lower()
                word = "".join(c for c in unicodedata.normalize("NFD", word) if unicodedata.category(c) != "Mn")
            piece: List[str] = []
            for ch in word:
                if _is_punctuation(ch):
                    if piece:
                        out.append("".join(piece))
                        piece = []
                    out.append(ch)
                else:
                    piece.append(ch)
            if piece:
                out.append("".join(piece))
        # a second whitespace split, as the reference does after re-joining (a token may have become empty / gained blanks)
        return " ".join(out).split()

    # -- stage 2 ---------------------------------------------------------------------------
    def _wordpiece(self, word: str) -> List[str]:
        if len(word) > MAX_INPUT_CHARS_PER_WORD:
            return [self.unk_token]
        pieces: List[str] = []
        start, n = 0, len(word)
        while start < n:
            end = n
            found = None
            while start < end:
                sub = word[start:end]
                if start > 0:
                    sub = "##" + sub
                if sub in self.vocab:
                    found = sub
                    break
                end -= 1
            if found is None:
                return [self.unk_token]
            pieces.append(found)
            start = end
        return pieces

    def tokenize(self, text: str) -> List[str]:
        out: List[str] = []
        for word in self._basic(text):
            out.extend(self._wordpiece(word))
        return out

    def convert_tokens_to_ids(self, tokens: Iterable[str]) -> List[int]:
        return [self.vocab[t] for t in tokens]

    def convert_ids_to_tokens(self, ids: Iterable[int]) -> List[str]:
        return [self.inv_vocab[int(i)] for i in ids]

    def tokenize_batch(self, texts: Union[str, List[str]], context_length: int = 32) -> torch.Tensor:
        """wukong_clip/data.py:181-203 / predictor.py:56-78"""
        if isinstance(texts, str):
            texts = [texts]
        cls, sep = self.vocab["[CLS]"], self.vocab["[SEP]"]
        result = torch.zeros(len(texts), context_length, dtype=torch.long)
        for i, text in enumerate(texts):
            ids = [cls] + self.convert_tokens_to_ids(self.tokenize(text))[:context_length - 2] + [sep]
            result[i, :len(ids)] = torch.tensor(ids)
        return result


def _read_vocab(vocab_file: str) -> Dict[str, int]:
    """one token per line, id = line number (a repeated token keeps its LAST line, as a dict assignment does)"""
    vocab: Dict[str, int] = {}
    index = 0
    with io.open(vocab_file, "r", encoding="utf-8") as f:
        while True:
            line = f.readline()
            if not line:
                break
            vocab[line.strip()] = index
            index += 1
    return vocab
