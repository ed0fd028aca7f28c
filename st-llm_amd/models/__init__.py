"""Host-side mirror of the reference's ``stllm.models`` surface (same module, class and parameter names)."""
from . import Qformer, blip2, eva_vit, llama, st_llm  # noqa: F401
from .st_llm import STLLMForCausalLM, STLLMLlamaModel, STLLMModel, StllmConfig  # noqa: F401
