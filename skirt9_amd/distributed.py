"""Multi-GPU sharding of a primary-emission segment for Python callers that hold the detector arrays in torch tensors:
one process per GPU, torch.distributed (backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests).  The product path
without torch is in the engine library itself: ``pmc_reduce_frames`` / ``pmc_allreduce_radiation_field`` over an RCCL
communicator (include/pmc.h; ``skirt_mi355x -g 0,1,...`` drives one host thread per device through them).

Ordering contract: the engine runs on its own HIP streams.  ``Engine.run_primary`` returns only when the segment is
complete on the device, so a collective issued on torch's stream afterwards sees the finished arrays; before the NEXT
``run_primary`` the caller must make torch's work on the bound tensors visible (``torch.cuda.synchronize()`` or an
event), as ``bench.py`` does.

Photon histories are independent (performLifeCycle touches only thread-local state and atomically-added detector
arrays), so a segment of Npp histories is split statically by index -- rank g of G takes
[floor(g*Npp/G), floor((g+1)*Npp/G)) -- replacing the reference's chunk server (MultiHybridParallel.cpp:26-104).
Every rank holds a full replica of grid, densities and tables.  The ONE exchange step is the sum of the detector
arrays onto rank 0 at the end of the segment, the counterpart of ProcessManager::sumToRoot
(FluxRecorder.cpp:487-493, ProcessManager.cpp:223-255).  The statistics arrays (sum of w^k per history) are additive
too because a history lives on exactly one rank.
"""


def history_range(num_packets, rank, world):
    """[first, first+count) of this rank for a segment of num_packets histories: pmc_history_range of the engine library
    (include/pmc.h), the implementation the CLI driver ``skirt_mi355x -g 0,1,...`` uses as well"""
    from .engine import history_range as _range
    return _range(num_packets, rank, world)


def reduce_frames(frames, dst=0):
    """sum the detector-array tensor of every rank onto rank dst (in place on dst); no-op for a single process"""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(frames, dst=dst, op=dist.ReduceOp.SUM)
    return frames


def allreduce_radiation_field(rf):
    """sum the radiation field table of every rank onto ALL ranks (in place): the counterpart of
    MediumSystem::communicateRadiationField (MediumSystem.cpp:1304-1313, ProcessManager::sumToAll) -- every process
    needs the whole field for what follows the primary segment (dust emission).  `rf` is a float64 tensor that aliases
    pmc_radiation_field_device() (or a gloo tensor in the CPU tests); no-op for a single process"""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(rf, op=dist.ReduceOp.SUM)
    return rf
