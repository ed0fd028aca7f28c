"""Token generation on the KV cache: what `Chat.answer` asks of `llama_model.generate(...)` in the reference
(stllm/conversation/conversation.py:231-243: inputs_embeds, max_new_tokens, stopping_criteria, num_beams, do_sample,
min_length, top_p, repetition_penalty, length_penalty, temperature; demo.py:58-66 runs it with num_beams=5, do_sample=False).

The reference delegates to HuggingFace transformers == 4.28.0 (requirement.txt), which is not part of /root/reference; this
module restates the published algorithm of `GenerationMixin.greedy_search / sample / beam_search` + `BeamSearchScorer` for
the argument subset above (one return sequence, no beam groups):

  * logits processors, in HF's order: RepetitionPenaltyLogitsProcessor, MinLengthLogitsProcessor; sampling adds the warpers
    TemperatureLogitsWarper, TopPLogitsWarper;
  * greedy / sampling: finished rows emit `pad_token_id`; stop when every row has emitted EOS, a stopping criterion fires, or
    `max_new_tokens` tokens exist;
  * beam search: scores are log-softmax + running beam score, top 2*num_beams candidates per step, candidates ending in EOS
    become hypotheses (score = sum_logprobs / len ** length_penalty, only if ranked inside the first num_beams), the rest
    refill the beams; a batch row is done when num_beams hypotheses exist and the best running score cannot beat the worst
    of them (early_stopping=False heuristic: best_sum_logprobs / cur_len ** length_penalty); `finalize` adds the open beams
    and returns the best hypothesis followed by EOS when it ended early;
  * the KV cache is re-ordered by `beam_idx` after every step (`LlamaForCausalLM._reorder_cache`, spec
    modeling_llama_mem.py:747-752) — here one index_select per layer on the fused QKV cache rows that are in use.

With `inputs_embeds` and no `input_ids` the sequences HF scores start empty, so every length below counts generated tokens
only.  The prompt is prefilled ONCE and its cache rows are replicated to the beams (HF prefills num_beams identical copies).
"""
import torch


class _BeamHypotheses:
    """transformers 4.28 generation/beam_search.py BeamHypotheses (early_stopping=False)."""

    def __init__(self, num_beams, length_penalty):
        self.num_beams, self.length_penalty = num_beams, length_penalty
        self.beams = []
        self.worst_score = 1e9

    def __len__(self):
        return len(self.beams)

    def add(self, hyp, sum_logprobs):
        score = sum_logprobs / (hyp.shape[-1] ** self.length_penalty)
        if len(self) < self.num_beams or score > self.worst_score:
            self.beams.append((score, hyp))
            if len(self) > self.num_beams:
                sorted_scores = sorted([(s, idx) for idx, (s, _) in enumerate(self.beams)])
                del self.beams[sorted_scores[0][1]]
                self.worst_score = sorted_scores[1][0]
            else:
                self.worst_score = min(score, self.worst_score)

    def is_done(self, best_sum_logprobs, cur_len):
        if len(self) < self.num_beams:
            return False
        return self.worst_score >= best_sum_logprobs / cur_len ** self.length_penalty


def _process_logits(scores, ids, cur_len, repetition_penalty, min_length, eos_token_id):
    """RepetitionPenaltyLogitsProcessor then MinLengthLogitsProcessor (HF order), on [rows, vocab] fp32 scores."""
    if repetition_penalty != 1.0 and ids.shape[1] > 0:
        s = torch.gather(scores, 1, ids)
        s = torch.where(s < 0, s * repetition_penalty, s / repetition_penalty)
        scores = scores.scatter(1, ids, s)
    if eos_token_id is not None and cur_len < min_length:
        scores = scores.clone()
        scores[:, eos_token_id] = -float("inf")
    return scores


def _warp_logits(scores, temperature, top_p):
    """TemperatureLogitsWarper then TopPLogitsWarper (min_tokens_to_keep = 1)."""
    if temperature != 1.0:
        scores = scores / temperature
    if top_p is not None and top_p < 1.0:
        sorted_logits, sorted_indices = torch.sort(scores, descending=False)
        cumulative = sorted_logits.softmax(dim=-1).cumsum(dim=-1)
        remove = cumulative <= (1 - top_p)
        remove[..., -1:] = False
        scores = scores.masked_fill(remove.scatter(1, sorted_indices, remove), -float("inf"))
    return scores


def _stop(stopping_criteria, ids, scores):
    return stopping_criteria is not None and any(bool(sc(ids, scores)) for sc in stopping_criteria)


class _CacheStepper:
    """next-token logits from the model: prefill once, then one decode step per token; rows can be re-ordered (beams)."""

    def __init__(self, lm_wrapper, inputs_embeds, rows, max_new_tokens, use_cache=True):
        self.m, self.lm = lm_wrapper, lm_wrapper.model
        self.use_cache = use_cache
        emb = inputs_embeds.float()
        B, S, _ = emb.shape
        self.rep = rows // B                      # beams per prompt
        if use_cache:
            self.cache = self.lm.new_cache(B, S + max_new_tokens, emb.device)
            _, h16 = self.lm.prefill(emb, None, cache=self.cache)
            logits = self.m.logits_from(h16.view(B, S, -1)[:, -1].contiguous(), B, 1)[:, 0]
            if self.rep > 1:                      # replicate the prompt's cache rows to the beams
                big = self.lm.new_cache(rows, S + max_new_tokens, emb.device)
                for dst, src in zip(big.qkv, self.cache.qkv):
                    dst[:, :S] = src[:, :S].repeat_interleave(self.rep, dim=0)
                big.len = self.cache.len
                self.cache = big
                logits = logits.repeat_interleave(self.rep, dim=0)
            self.logits = logits.float()
        else:
            self.emb = emb.repeat_interleave(self.rep, dim=0)
            self.logits = self.m.forward(samples=None, inputs_embeds=self.emb).logits[:, -1].float()

    def advance(self, next_tokens, beam_idx=None):
        """append `next_tokens` [rows] (after re-ordering the rows by `beam_idx`) and compute the next logits"""
        tok = self.lm.embed_tokens(next_tokens.view(-1, 1).cpu())
        if self.use_cache:
            if beam_idx is not None and not torch.equal(beam_idx.cpu(), torch.arange(beam_idx.numel())):
                n = self.cache.len
                idx = beam_idx.to(self.cache.qkv[0].device)
                for buf in self.cache.qkv:        # _reorder_cache: past.index_select(0, beam_idx), only the rows in use
                    buf[:, :n] = buf[:, :n].index_select(0, idx)
            _, h16 = self.lm.decode_step(tok, self.cache)
            self.logits = self.m.logits_from(h16, next_tokens.numel(), 1)[:, 0].float()
        else:
            if beam_idx is not None:
                self.emb = self.emb.index_select(0, beam_idx.to(self.emb.device))
            self.emb = torch.cat([self.emb, tok.to(self.emb.device)], dim=1)
            self.logits = self.m.forward(samples=None, inputs_embeds=self.emb).logits[:, -1].float()


@torch.no_grad()
def generate(lm_wrapper, inputs_embeds, max_new_tokens=16, num_beams=1, do_sample=False, min_length=0, top_p=1.0,
             temperature=1.0, repetition_penalty=1.0, length_penalty=1.0, stopping_criteria=None, eos_token_id=2,
             pad_token_id=0, use_cache=True, generator=None):
    """Returns the generated ids [B, n] (the prompt has no ids), HF semantics as described in the module docstring."""
    B = inputs_embeds.shape[0]
    dev = inputs_embeds.device
    if num_beams == 1:
        st = _CacheStepper(lm_wrapper, inputs_embeds, B, max_new_tokens, use_cache)
        ids = torch.zeros((B, 0), dtype=torch.long, device=dev)
        unfinished = torch.ones(B, dtype=torch.long, device=dev)
        while True:
            scores = _process_logits(st.logits.to(dev), ids, ids.shape[1], repetition_penalty, min_length, eos_token_id)
            if do_sample:
                probs = _warp_logits(scores, temperature, top_p).softmax(dim=-1)
                nxt = torch.multinomial(probs, num_samples=1, generator=generator).squeeze(1)
            else:
                nxt = scores.argmax(dim=-1)
            if eos_token_id is not None:
                nxt = nxt * unfinished + pad_token_id * (1 - unfinished)
            ids = torch.cat([ids, nxt[:, None]], dim=-1)
            if eos_token_id is not None:
                unfinished = unfinished * (nxt != eos_token_id).long()
            if unfinished.max() == 0 or _stop(stopping_criteria, ids, scores) or ids.shape[1] >= max_new_tokens:
                break
            st.advance(nxt)
        _check_exchanges(dev)
        return ids

    if do_sample:
        raise NotImplementedError("beam-sample is not used by the reference (demo: num_beams=5, do_sample=False)")
    nb = num_beams
    st = _CacheStepper(lm_wrapper, inputs_embeds, B * nb, max_new_tokens, use_cache)
    ids = torch.zeros((B * nb, 0), dtype=torch.long, device=dev)
    beam_scores = torch.zeros((B, nb), dtype=torch.float32, device=dev)
    beam_scores[:, 1:] = -1e9
    beam_scores = beam_scores.view(-1)
    hyps = [_BeamHypotheses(nb, length_penalty) for _ in range(B)]
    done = [False] * B
    while True:
        cur_len = ids.shape[1]
        logp = st.logits.to(dev).log_softmax(dim=-1)
        logp = _process_logits(logp, ids, cur_len, repetition_penalty, min_length, eos_token_id)
        V = logp.shape[-1]
        cand = (logp + beam_scores[:, None]).view(B, nb * V)
        cand_scores, cand_tokens = torch.topk(cand, 2 * nb, dim=1, largest=True, sorted=True)
        cand_beams = torch.div(cand_tokens, V, rounding_mode="floor")
        cand_tokens = cand_tokens % V
        next_scores = torch.zeros((B, nb), dtype=torch.float32, device=dev)
        next_tokens = torch.zeros((B, nb), dtype=torch.long, device=dev)
        next_index = torch.zeros((B, nb), dtype=torch.long, device=dev)
        cs, ct, cb = cand_scores.cpu(), cand_tokens.cpu(), cand_beams.cpu()
        for b in range(B):                                   # BeamSearchScorer.process
            if done[b]:
                next_scores[b] = 0
                next_tokens[b] = pad_token_id
                next_index[b] = b * nb
                continue
            k = 0
            for rank in range(2 * nb):
                tok, sc, row = int(ct[b, rank]), float(cs[b, rank]), b * nb + int(cb[b, rank])
                if eos_token_id is not None and tok == eos_token_id:
                    if rank >= nb:
                        continue
                    hyps[b].add(ids[row].clone(), sc)
                else:
                    next_scores[b, k], next_tokens[b, k], next_index[b, k] = sc, tok, row
                    k += 1
                if k == nb:
                    break
            done[b] = done[b] or hyps[b].is_done(float(cs[b].max()), cur_len)
        beam_scores = next_scores.view(-1)
        beam_idx = next_index.view(-1)
        ids = torch.cat([ids[beam_idx], next_tokens.view(-1, 1)], dim=-1)
        if all(done) or _stop(stopping_criteria, ids, None) or ids.shape[1] >= max_new_tokens:
            break
        st.advance(next_tokens.view(-1), beam_idx)
    # BeamSearchScorer.finalize: open beams become hypotheses, best one wins, EOS appended when it ended early
    out = []
    for b in range(B):
        if not done[b]:
            for j in range(nb):
                hyps[b].add(ids[b * nb + j], float(beam_scores[b * nb + j]))
        out.append(sorted(hyps[b].beams, key=lambda x: x[0])[-1][1])
    sent_max = min(max(len(o) for o in out) + 1, max_new_tokens)
    res = torch.full((B, sent_max), pad_token_id, dtype=torch.long, device=dev)
    for b, o in enumerate(out):
        res[b, :len(o)] = o
        if len(o) < sent_max:
            res[b, len(o)] = eos_token_id
    _check_exchanges(dev)
    return res


def _check_exchanges(dev):
    """every token was read back on the host, so the stream is idle: a timed-out split-K exchange anywhere in this generate()
    call must not come back as plausible-looking ids"""
    if torch.device(dev).type == "cuda":
        from . import hip
        hip.gemm_workspace_check(dev, wait=True)
