"""Caption tokenisation (host side, integer-exact).

`load_tokenizer(path)` is the loader of /root/reference/tokenizer/tokenizer.py:4-9 (a `tokenizer.json`
trained by tokenizer/train_tokenizer.py: BPE, Whitespace pre-tokeniser, specials [PAD]=0 [CLS]=1 [SEP]=2
[UNK]=3, template `[CLS] $A [SEP]`).  The reference ships no tokenizer file (its .gitignore excludes
*.json) and its generator downloads a pretrained one, which is impossible offline, so `HashTokenizer`
provides the same call convention without any file: identical special ids, template and Whitespace
pre-tokenisation, token ids by a stable hash.
"""
from __future__ import annotations

import re
import zlib
from types import SimpleNamespace

import torch

PAD, CLS, SEP, UNK = 0, 1, 2, 3
_PRETOK = re.compile(r"\w+|[^\w\s]+")


def load_tokenizer(tokenizer_file):
    from transformers import PreTrainedTokenizerFast
    tok = PreTrainedTokenizerFast(tokenizer_file=tokenizer_file)
    tok.pad_token = "[PAD]"
    tok.cls_token = "[CLS]"
    tok.sep_token = "[SEP]"
    return tok


class _Encoding(SimpleNamespace):
    def to(self, device):
        self.input_ids = self.input_ids.to(device)
        self.attention_mask = self.attention_mask.to(device)
        return self


class HashTokenizer:
    """File-free tokenizer with the reference call convention: tok(texts, padding=..., max_length=...,
    truncation=True, return_tensors="pt") -> .input_ids, .attention_mask."""

    pad_token_id, cls_token_id, sep_token_id, unk_token_id = PAD, CLS, SEP, UNK

    def __init__(self, vocab_size=16384):
        self.vocab_size = vocab_size

    def _ids(self, text):
        return [4 + zlib.crc32(t.encode("utf-8")) % (self.vocab_size - 4) for t in _PRETOK.findall(text)]

    def __call__(self, text, padding=True, max_length=77, truncation=True, return_tensors="pt"):
        if isinstance(text, str):
            text = [text]
        seqs = []
        for t in text:
            ids = [CLS] + self._ids(t) + [SEP]
            if truncation and len(ids) > max_length:
                ids = ids[:max_length - 1] + [SEP]
            seqs.append(ids)
        width = max_length if padding == "max_length" else max(len(s) for s in seqs)
        ids = torch.full((len(seqs), width), PAD, dtype=torch.long)
        mask = torch.zeros((len(seqs), width), dtype=torch.long)
        for i, s in enumerate(seqs):
            ids[i, :len(s)] = torch.tensor(s, dtype=torch.long)
            mask[i, :len(s)] = 1
        return _Encoding(input_ids=ids, attention_mask=mask)
