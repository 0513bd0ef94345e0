"""dpr_scale_amd -- MI355X-native in-batch contrastive hot path of dpr-scale.

Public surface:
  dpr_scale_amd.hotpath.InBatchContrastive / inbatch_contrastive_loss   the operator (autograd)
  dpr_scale_amd.hotpath.sim_score / cross_entropy_mean / rank_of_gold / topk   forward-only pieces
  dpr_scale_amd.task.dpr_task.DenseRetrieverTask   drop-in for dpr_scale.task.dpr_task.DenseRetrieverTask
The HIP library (libdprhot.so) is loaded on first use; there is no CPU fallback.
"""
__version__ = "0.1.0"
