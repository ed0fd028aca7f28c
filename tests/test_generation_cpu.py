"""CPU: the decode-loop bookkeeping (stllm_amd/generation.py: greedy, beam search with KV-cache re-ordering, repetition
penalty, min_length, stopping) against HuggingFace transformers' own `generate` — the implementation the reference delegates
to (conversation.py:231-243) — on a small random Llama (head_dim 128) whose weights are shared between the two.
The product model runs on the test-only CPU backend (tests/_cpu_backend.py), so only host logic is under test here."""
import pytest
import torch

import _cpu_backend

transformers = pytest.importorskip("transformers")

HID, INTER, LAYERS, HEADS, VOCAB = 256, 512, 2, 2, 256


def _models(seed):
    from transformers import LlamaConfig, LlamaForCausalLM
    from stllm_amd import synth
    from stllm_amd.models.st_llm import STLLMForCausalLM, StllmConfig
    mine = STLLMForCausalLM(StllmConfig(hidden_size=HID, intermediate_size=INTER, num_hidden_layers=LAYERS, num_attention_heads=HEADS,
                                        vocab_size=VOCAB, rms_norm_eps=1e-6), device="cpu")
    synth.fill_module_(mine, seed, "")
    with torch.no_grad():   # a livelier next-token distribution than N(0, 0.02) weights give
        mine.lm_head.weight.mul_(40.0)
    hf = LlamaForCausalLM(LlamaConfig(hidden_size=HID, intermediate_size=INTER, num_hidden_layers=LAYERS, num_attention_heads=HEADS,
                                      num_key_value_heads=HEADS, vocab_size=VOCAB, rms_norm_eps=1e-6, max_position_embeddings=512,
                                      rope_theta=10000.0, attention_bias=False, tie_word_embeddings=False)).eval()
    missing, unexpected = hf.load_state_dict({k: v for k, v in mine.state_dict().items()}, strict=False)
    assert not [m for m in missing if "rotary" not in m] and not unexpected, (missing, unexpected)
    return mine, hf


def _run_mine(mine, emb, **kw):
    from stllm_amd import runtime
    with _cpu_backend.installed(), runtime.use_dtype("fp32"):
        return mine.generate(inputs_embeds=emb, **kw)


def _strip(t, pad=0):
    return [int(x) for x in t]


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_greedy_matches_hf(seed):
    mine, hf = _models(seed)
    torch.manual_seed(seed)
    emb = torch.randn(2, 6, HID) * 0.5
    kw = dict(max_new_tokens=8, do_sample=False, num_beams=1, min_length=2, repetition_penalty=1.3)
    got = _run_mine(mine, emb, **kw)
    ref = hf.generate(inputs_embeds=emb, pad_token_id=0, eos_token_id=2, **kw)
    assert got.shape == ref.shape and torch.equal(got, ref), (got, ref)
    # the KV-cache path and the re-prefill path agree
    assert torch.equal(got, _run_mine(mine, emb, use_cache=False, **kw))


@pytest.mark.parametrize("seed,beams,lp,rp", [(0, 3, 1.0, 1.0), (1, 5, 1.0, 1.0), (2, 4, 2.0, 1.2), (3, 2, 0.5, 1.0), (4, 5, 1.0, 1.1)])
def test_beam_search_matches_hf(seed, beams, lp, rp):
    """demo.py:58-66 decodes with num_beams=5, do_sample=False"""
    mine, hf = _models(seed)
    torch.manual_seed(100 + seed)
    emb = torch.randn(1, 5, HID) * 0.5
    kw = dict(max_new_tokens=7, do_sample=False, num_beams=beams, min_length=1, repetition_penalty=rp, length_penalty=lp)
    got = _run_mine(mine, emb, **kw)
    ref = hf.generate(inputs_embeds=emb, pad_token_id=0, eos_token_id=2, early_stopping=False, **kw)
    assert _strip(got[0]) == _strip(ref[0]), (got, ref)
    assert torch.equal(got, _run_mine(mine, emb, use_cache=False, **kw)), "KV-cache re-ordering differs from re-prefilling the beams"


def test_eos_and_stopping_criteria():
    mine, hf = _models(7)
    torch.manual_seed(7)
    emb = torch.randn(1, 4, HID) * 0.5
    free = _run_mine(mine, emb, max_new_tokens=6)
    # make a later generated token the EOS (the first one, from position 1 on, that has not occurred before): generation stops with it
    seq = free[0].tolist()
    j = next(i for i in range(1, len(seq)) if seq[i] not in seq[:i])
    eos = seq[j]
    got = _run_mine(mine, emb, max_new_tokens=6, eos_token_id=eos)
    assert got.tolist() == [seq[:j + 1]]
    ref = hf.generate(inputs_embeds=emb, max_new_tokens=6, do_sample=False, pad_token_id=0, eos_token_id=eos)
    assert got.tolist() == ref.tolist()

    class StopOn:   # conversation.py:168-178 StoppingCriteriaSub: stop when the tail of the first row equals a stop sequence
        def __init__(self, stops):
            self.stops = stops

        def __call__(self, input_ids, scores):
            return any(len(input_ids[0]) >= len(s) and torch.equal(input_ids[0][-len(s):], s) for s in self.stops)

    stop = free[0, j - 1:j + 1].clone()          # the two-token tail ending at position j
    got = _run_mine(mine, emb, max_new_tokens=6, stopping_criteria=[StopOn([stop])])
    first = next(i for i in range(1, len(seq)) if seq[i - 1:i + 1] == stop.tolist())
    assert got.tolist() == [seq[:first + 1]]


def test_sampling_is_seeded_and_top_p_restricts():
    mine, _ = _models(9)
    torch.manual_seed(9)
    emb = torch.randn(1, 4, HID) * 0.5
    g = lambda: torch.Generator().manual_seed(123)
    a = _run_mine(mine, emb, max_new_tokens=5, do_sample=True, top_p=0.9, temperature=0.8, generator=g())
    b = _run_mine(mine, emb, max_new_tokens=5, do_sample=True, top_p=0.9, temperature=0.8, generator=g())
    assert torch.equal(a, b)
    greedy = _run_mine(mine, emb, max_new_tokens=5)
    tiny_p = _run_mine(mine, emb, max_new_tokens=5, do_sample=True, top_p=1e-6, generator=g())   # nucleus of one token == greedy
    assert torch.equal(tiny_p, greedy)


def test_stopping_criteria_sub_semantics():
    """conversation.py:105-116, including its behaviour on rows shorter than a stop sequence (broadcast compare)"""
    from stllm_amd.conversation import StoppingCriteriaSub
    sc = StoppingCriteriaSub(stops=[torch.tensor([835]), torch.tensor([2277, 29937])])
    assert sc(torch.tensor([[5, 835]]), None) and sc(torch.tensor([[1, 2277, 29937], [0, 0, 0]]), None)
    assert not sc(torch.tensor([[835, 5]]), None) and not sc(torch.tensor([[2277]]), None)
    assert not sc(torch.tensor([[7, 7], [5, 835]]), None)          # only the first row is looked at
