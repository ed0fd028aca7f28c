"""Offline stand-in for the reference's HF tokenizers (LlamaTokenizer / BertTokenizer are host-side
text utilities, out of scope; no vocab files exist offline — SURVEY.md §8a row A0/A14).

A "text" is a string of whitespace-separated integer token ids; any other word (role tags such as
``Human:`` / ``###``) carries no ids.  The call signature mirrors the subset of the HF tokenizer API the
reference uses (st_llm.py:344-350, 387-390, 501-508; conversation.py:327-328), so a real HF tokenizer
object can be dropped in instead when its files are available.
"""
import types

import torch


class IdTokenizer:
    def __init__(self, pad_token_id=0, bos_token_id=1, eos_token_id=2, vocab_size=32000):
        self.pad_token_id, self.bos_token_id, self.eos_token_id = pad_token_id, bos_token_id, eos_token_id
        self.eos_token = f" {eos_token_id}"
        self.pad_token = None
        self.padding_side = "right"
        self.vocab_size = vocab_size

    def __len__(self):
        return self.vocab_size

    def add_special_tokens(self, d):
        return 0

    def encode_ids(self, s, add_special_tokens=True):
        ids = [int(t) for t in s.split() if t.isdigit()]
        return ([self.bos_token_id] + ids) if add_special_tokens else ids

    def __call__(self, text, return_tensors="pt", add_special_tokens=True, padding=None, truncation=False,
                 max_length=None, **kw):
        texts = [text] if isinstance(text, str) else list(text)
        rows = [self.encode_ids(t, add_special_tokens) for t in texts]
        if truncation and max_length is not None:
            rows = [r[:max_length] for r in rows]
        L = max((len(r) for r in rows), default=0)
        ids = torch.full((len(rows), L), self.pad_token_id, dtype=torch.long)
        att = torch.zeros((len(rows), L), dtype=torch.long)
        for i, r in enumerate(rows):
            ids[i, :len(r)] = torch.tensor(r, dtype=torch.long)
            att[i, :len(r)] = 1
        out = types.SimpleNamespace(input_ids=ids, attention_mask=att)
        out.to = lambda dev: out  # ids stay on the host: assembly builds gather indices there
        return out

    def decode(self, ids, **kw):
        return " ".join(str(int(i)) for i in ids)
