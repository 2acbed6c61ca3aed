"""When do the data-parallel collectives of the training path run?

Normally: a ``torch.distributed`` process group with more than one rank (one process per GPU under torchrun; backend "nccl" IS
RCCL on ROCm).  ``EML_DIST_SINGLE=1`` is the single-GPU dry run of that path: the entry points then
initialise the process group with ONE rank too, wrap the networks in DistributedDataParallel and issue every collective of a
multi-rank iteration (gradient buckets, SPADE's synchronised BatchNorm sums, the Sinkhorn diameter) through RCCL on the one
device -- same results as without it (a one-rank all-reduce is the identity), but RCCL's initialisation, its kernels on the
HIP stream and DDP's hooks have then run under this code before an 8-GPU node ever sees it."""
import os


def single_rank_dry_run():
    return os.environ.get("EML_DIST_SINGLE") == "1"


def dp_wrap(world):
    """Should a trainer built for ``world`` ranks wrap its networks in DistributedDataParallel?"""
    if world > 1:
        return True
    import torch.distributed as dist
    return single_rank_dry_run() and dist.is_available() and dist.is_initialized()


def dp_active():
    """Should a collective of the data-parallel path (sync-BN sums, global diameter) be issued now?"""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or single_rank_dry_run()
