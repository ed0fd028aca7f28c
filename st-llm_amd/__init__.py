"""stllm_amd — MI355X-native (gfx950) implementation of ST-LLM's video-token hot path.

Host side mirrors the reference's ``stllm.models`` surface (``stllm_amd.models.{eva_vit,
Qformer,blip2,st_llm}``); all arithmetic runs in hand-written HIP kernels behind the C ABI
declared in ``include/stllm_hip.h`` (``stllm_amd.hip`` is the ctypes binding).
"""
__version__ = "0.1.0"
