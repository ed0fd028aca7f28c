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
    # True: add_special_tokens({'pad_token': '[PAD]'}) behaves like HF's LlamaTokenizer — the new token gets the id
    # len(tokenizer) and the vocabulary grows by one (Vicuna: id 32000, 32001 words), which is what the reference's
    # InstructBLIP-style models are trained with (st_llm.py:306-310, :52-53, :180-181).  False: a no-op, like the fake tokenizer
    # the golden fixtures were generated with (tests/golden/ref_shim.py) — the fixture-replay tests switch it off.
    hf_special_tokens = True

    def __init__(self, pad_token_id=0, bos_token_id=1, eos_token_id=2, vocab_size=32000):
        self.pad_token_id, self.bos_token_id, self.eos_token_id = pad_token_id, bos_token_id, eos_token_id
        self.eos_token = f" {eos_token_id}"
        self.pad_token = None
        self.padding_side = "right"
        self.vocab_size = vocab_size

    def __len__(self):
        return self.vocab_size

    def add_special_tokens(self, d):
        """HF semantics for the one token the reference adds that is not in the vocabulary yet ('[PAD]'); '</s>' (bos / eos /
        unk in st_llm.py:308-310) already has an id.  Returns the number of tokens added."""
        if not self.hf_special_tokens or "pad_token" not in d or self.pad_token is not None:
            return 0
        self.pad_token = d["pad_token"]
        self.pad_token_id = self.vocab_size
        self.vocab_size += 1
        return 1

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
