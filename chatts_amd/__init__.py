"""chatts_amd - MI355X-native (gfx950) ChatTS inference hot path.

  ChatTSProcessor     processor(text=..., timeseries=...)        (AutoProcessor surface)
  ChatTSForCausalLM   model.generate(**inputs, max_new_tokens=)  (AutoModelForCausalLM surface + vLLM plugin hooks)
  LLM, SamplingParams vLLM-style offline engine look-alike
  PeftModel           PeftModel.from_pretrained(base, adapter_dir): LoRA adapter merged at load (demo/demo_lora.ipynb)
All device arithmetic lives in lib/libchatts_amd.so (hand-written HIP behind the C-ABI of include/chatts_amd.h).
"""
from .config import ChatTSConfig, preset  # noqa: F401
from .processing import ChatTSProcessor  # noqa: F401
from .tokenizer import SyntheticTokenizer  # noqa: F401

__all__ = ["ChatTSConfig", "preset", "ChatTSProcessor", "SyntheticTokenizer", "ChatTSForCausalLM", "LLM", "SamplingParams",
           "PeftModel"]


def __getattr__(name):          # torch-dependent pieces are imported lazily
    if name == "ChatTSForCausalLM":
        from .modeling import ChatTSForCausalLM
        return ChatTSForCausalLM
    if name in ("LLM", "SamplingParams"):
        from . import llm
        return getattr(llm, name)
    if name == "PeftModel":
        from .lora import PeftModel
        return PeftModel
    raise AttributeError(name)
