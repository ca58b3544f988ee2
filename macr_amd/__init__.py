"""macr_amd -- MI355X (gfx950) native hot path of MACR.

Host side: Python on PyTorch-ROCm tensors (device memory, streams,
torch.distributed/RCCL are plumbing).  Device side: hand-written HIP kernels
behind the C ABI of include/macr_hip.h (macr_amd/csrc -> libmacr_hip.so).
There is NO CPU or PyTorch fallback for the kernels: importing macr_amd.ops
without the built extension raises.
"""
__version__ = "0.1.0"
