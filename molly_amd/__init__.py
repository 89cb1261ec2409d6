"""molly_amd — MI355X-native hot path of the Molly multi-omics LLM."""
from .config import EncConfig, LlmConfig, OmicsModalConfig, get_omics_one_config  # noqa: F401


def __getattr__(name):
    # model classes import torch.nn lazily so that `import molly_amd` stays cheap for the ABI tests
    if name in ("OmicsOne", "Qwen3ForCausalLM", "EsmForMaskedLM", "CausalLMOutputWithPast"):
        from . import model
        return getattr(model, name)
    raise AttributeError(name)
