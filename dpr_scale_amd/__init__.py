"""dpr_scale_amd -- MI355X-native in-batch contrastive hot path of dpr-scale.

Public surface:
  dpr_scale_amd.hotpath.InBatchContrastive / inbatch_contrastive_loss   the operator (autograd)
  dpr_scale_amd.hotpath.sim_score / cross_entropy_mean / rank_of_gold / topk   forward-only pieces
  dpr_scale_amd.task.dpr_task.DenseRetrieverTask   drop-in for dpr_scale.task.dpr_task.DenseRetrieverTask
The HIP library (libdprhot.so) is loaded on first use; there is no CPU fallback.
"""
import os as _os

_RUNTIME = {"done": False}


def configure_runtime(dev_kernarg=True, nccl_high_priority=True, warn=True):
    """Two process-wide runtime switches the hot path was measured with.  EXPLICIT since round 6: importing the package no
    longer touches the environment; bench.py, the scripts and DenseRetrieverTask.__init__ call this (DPRHOT_RUNTIME_DEFAULTS=0 makes it a
    no-op; an explicit value already in the environment always wins).  Both are read ONCE by their runtime, so each is reported -- not
    silently ignored -- when that runtime is already up:
      HIP_FORCE_DEV_KERNARG=1     kernel arguments in device memory, read when the HIP runtime initialises.  The path's launches are a
                                  few microseconds each and every one begins by loading its arguments: from host-coherent memory that
                                  is a trip across PCIe per launch (cfg3-per-rank step, eager: 31.0 -> 26.0 us; bert-base step -1.7 %).
      TORCH_NCCL_HIGH_PRIORITY=1  torch.distributed's RCCL streams at high priority, read when a process group is created -- it raises
                                  EVERY NCCL stream of the process (DDP's bucket all-reduces included).  HIP multiplexes a process's
                                  streams onto GPU_MAX_HW_QUEUES = 4 hardware queues and two streams that share one run in order: on
                                  the test box ProcessGroupNCCL's stream landed on the compute stream's queue and the "overlapped"
                                  collectives ran with the compute stream idle (profiles/r05_overlap_busy_torch_distributed*.json); a
                                  priority is a property of the hardware queue, so a high-priority stream cannot alias the compute
                                  stream's.
    Returns {name: "set" | "kept <value>" | "too late"}."""
    out = {}
    if _os.environ.get("DPRHOT_RUNTIME_DEFAULTS", "1") == "0":
        return out
    import sys as _sys
    import warnings as _warnings

    def late_hip():
        t = _sys.modules.get("torch")
        return bool(t is not None and t.cuda.is_initialized())

    def late_pg():
        t = _sys.modules.get("torch")
        try:
            return bool(t is not None and t.distributed.is_available() and t.distributed.is_initialized())
        except Exception:
            return False

    for want, name, late in ((dev_kernarg, "HIP_FORCE_DEV_KERNARG", late_hip), (nccl_high_priority, "TORCH_NCCL_HIGH_PRIORITY", late_pg)):
        if not want:
            continue
        if name in _os.environ:
            out[name] = "kept " + _os.environ[name]
        elif late():
            out[name] = "too late"
            if warn and not _RUNTIME["done"]:
                _warnings.warn(f"dpr_scale_amd.configure_runtime: {name} has no effect any more (its runtime is already initialised); "
                               "set it in the launcher's environment or call configure_runtime() earlier", RuntimeWarning, stacklevel=2)
        else:
            _os.environ[name] = "1"
            out[name] = "set"
    _RUNTIME["done"] = True
    return out


__version__ = "0.1.0"
