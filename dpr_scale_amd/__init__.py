"""dpr_scale_amd -- MI355X-native in-batch contrastive hot path of dpr-scale.

Public surface:
  dpr_scale_amd.hotpath.InBatchContrastive / inbatch_contrastive_loss   the operator (autograd)
  dpr_scale_amd.hotpath.sim_score / cross_entropy_mean / rank_of_gold / topk   forward-only pieces
  dpr_scale_amd.task.dpr_task.DenseRetrieverTask   drop-in for dpr_scale.task.dpr_task.DenseRetrieverTask
The HIP library (libdprhot.so) is loaded on first use; there is no CPU fallback.
"""
import os as _os

# Kernel arguments in device memory (ROCm runtime switch HIP_FORCE_DEV_KERNARG, read when the HIP runtime initialises -- i.e. it
# takes effect when this package is imported before the process's first HIP call; an explicit value in the environment wins).
# The hot path's launches are a few microseconds each and every one begins by loading its arguments: from host-coherent memory
# that is a trip across PCIe per launch (cfg3-per-rank step, eager: 31.0 -> 26.0 us with the arguments on the device).
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
# torch.distributed's RCCL streams at HIGH priority (read when a process group is created).  HIP multiplexes a process's streams onto
# GPU_MAX_HW_QUEUES = 4 hardware queues, and two streams that share one run in order: on the test box ProcessGroupNCCL's stream landed
# on the compute stream's queue, and the "overlapped" all-gather / reduce-scatter of the multi-GPU step ran with the compute stream idle
# (profiles/r05_overlap_busy_torch_distributed*.json).  A priority is a property of the hardware queue, so a high-priority stream
# cannot alias the (normal-priority) compute stream.
_os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")

__version__ = "0.1.0"
