"""crab_amd -- MI355X (gfx950) native forward path for GeWu-Lab/Crab's multimodal inference stack.

Layout: csrc/ (hand-written HIP kernels + the C-ABI, built to libcrab_hip.so), and the Python host side that
mirrors the reference's module boundaries (unified_llama / unified_qwen / unified_arch / multimodal_encoder /
peft_hyper) and drives the kernels through ctypes.  There is no CPU or eager-PyTorch compute fallback.
"""
__version__ = "0.1.0"
