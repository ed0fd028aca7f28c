"""Tensor path of the reference's Chat wrapper (stllm/conversation/conversation.py:181-340): what
``demo.py`` drives — ``upload_video`` -> ``encode_img`` -> pooling -> prompt-embedding concat ->
``generate(inputs_embeds=...)``.  The frame transform (resize / crop / normalise, conversation.py:190-198) runs on the
GPU (processors.VideoTransform); video decoding, prompt templates and stopping criteria are
host-side text/media utilities and out of scope."""
import torch

from . import hip
from .models.st_llm import get_residual_index
from .processors import VideoTransform, is_raw_frames


class StoppingCriteriaSub:
    """conversation.py:105-116: stop as soon as the FIRST row ends in one of the stop sequences ('###' has two encodings)."""

    def __init__(self, stops=(), encounters=1):
        self.stops = list(stops)

    def __call__(self, input_ids, scores):
        for stop in self.stops:
            if torch.all((stop.to(input_ids.device) == input_ids[0][-len(stop):])).item():
                return True
        return False


class Chat:
    def __init__(self, model, device="cuda:0"):
        self.device = device
        self.LLM = model
        # conversation.py:185-190 — the visual front-end hangs off model.model (or model.model.model under peft)
        self.model = model.model.stllm_model if hasattr(model.model, "stllm_model") else model.model.model.stllm_model
        self.transform = VideoTransform(device)   # conversation.py:190-198
        # conversation.py:199-201: '###' ends an answer
        self.stopping_criteria = [StoppingCriteriaSub(stops=[torch.tensor([835]), torch.tensor([2277, 29937])])]

    def upload_video(self, video, conv, img_list, num_frame=64, text=None):
        """conversation.py:274-299.  `video`: decoded raw frames (uint8 RGB [T,H,W,3] / list of PIL images — what the
        reference's load_video returns; transformed on the GPU by processors.VideoTransform == self.transform) or an already
        transformed frames tensor ([T*3,224,224] or [T,3,224,224], CLIP-normalised)."""
        if is_raw_frames(video):
            video = self.transform(video)
        frames = video.to(self.device)
        if frames.dim() == 3:
            bt, w, h = frames.shape
            frames = frames.view(bt // 3, 3, w, h)
        m = self.model
        video_emb, _, _ = m.encode_img(frames, text=text)  # [T,32,4096]
        if m.video_input == "mean":
            video_emb = hip.mean_t(video_emb.unsqueeze(0).contiguous())
        elif m.video_input == "all":
            video_emb = video_emb.reshape(1, -1, video_emb.shape[-1])
        elif m.video_input == "residual":
            video_emb = m.pool_video(video_emb.unsqueeze(0))[:, 0]
        img_list.append(video_emb)
        if conv is not None:
            conv.append_message(conv.roles[0], "<Video><ImageHere></Video>")
        return "Received."

    def get_context_emb_ids(self, img_list, question_ids):
        """conversation.py:322-340 (get_context_emb_sim) on token ids: cat(video_emb, embed([BOS]+question))."""
        tk = self.model.llama_tokenizer
        ids = [[tk.bos_token_id] + list(question_ids)]
        seg = self.model.embed_tokens(torch.tensor(ids))
        mixed = torch.cat((img_list[0], seg), dim=1)
        att = torch.ones(mixed.shape[:-1], dtype=torch.long, device=mixed.device)
        return mixed, att

    def get_context_emb_sim(self, conv, img_list, system=True):
        question = conv.messages[0][1].split("</Video> ")[1]
        question = (conv.system if system else "") + "###Human: " + question + " ###Assistant: "
        return self.get_context_emb_ids(img_list, self.model.llama_tokenizer.encode_ids(question, add_special_tokens=False))

    def answer(self, img_list, question_ids, max_new_tokens=300, num_beams=1, min_length=1, top_p=0.9,
               repetition_penalty=1.0, length_penalty=1, temperature=1.0, max_length=2000, do_sample=False,
               stopping_criteria=None, instruction=False, **kw):
        """conversation.py:213-253: keep the last `max_length - max_new_tokens` embeddings, generate with the reference's
        knobs (demo.py: num_beams=5, do_sample=False), drop a leading <unk> (0) / <s> (1) token.  This entry point is the
        `get_context_emb_sim` path (no conv.instruction: video tokens + question), for which the reference OVERRIDES
        repetition_penalty with 1.5 (conversation.py:219-220) whatever the caller passed; instruction=True keeps the argument."""
        embs, att = self.get_context_emb_ids(img_list, question_ids)
        if not instruction:
            repetition_penalty = 1.5
        begin = max(0, embs.shape[1] - (max_length - max_new_tokens))
        embs = embs[:, begin:]
        if stopping_criteria is None:
            stopping_criteria = self.stopping_criteria
        out = self.LLM.generate(inputs_embeds=embs, max_new_tokens=max_new_tokens, stopping_criteria=stopping_criteria,
                                num_beams=num_beams, do_sample=do_sample, min_length=min_length, top_p=top_p,
                                repetition_penalty=repetition_penalty, length_penalty=length_penalty, temperature=temperature, **kw)
        hip.gemm_workspace_check(embs.device, wait=True) if embs.is_cuda else None   # generate() synchronised on every token anyway
        tok = out[0]
        if tok.numel() and int(tok[0]) == 0:   # conversation.py:246-249
            tok = tok[1:]
        if tok.numel() and int(tok[0]) == 1:
            tok = tok[1:]
        return self.model.llama_tokenizer.decode(tok.tolist()), tok.cpu().numpy()
