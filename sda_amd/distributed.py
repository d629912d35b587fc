"""Multi-GPU form of the hot path: participants are sharded across ranks - one
process per GPU, no data-path collective while share vectors are generated and combined - and the
per-clerk partial sums meet in ONE exchange step at the end:

    direct reduce-scatter over the xGMI mesh  (all_to_all of 1/G slices: every pair of GPUs uses its
    own link, all 7 links busy, instead of a ring that is bound by one link)
      -> local modular sum of the G received slices (HIP kernel, exact 128-bit accumulation)
      -> all_gather of the reduced slices.

A plain `all_reduce(SUM)` on int64 is NOT usable: 8 residues of a 62-bit modulus overflow 2^64
(4 * q < 2^64 <= 8 * q for the 62-bit prime).

torch.distributed is plumbing here (RCCL over xGMI with backend "nccl"; gloo on CPU for tests).
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_participants(total: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous shard [first, first + count) of `total` participants for `rank`."""
    base, extra = divmod(total, world_size)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def _hip_modsum_parts(parts: torch.Tensor, modulus: int) -> torch.Tensor:
    """parts [G][len] int64 on the GPU -> [len] = column sum mod modulus (libsda_hip.so)."""
    from . import capi
    if not parts.is_cuda:
        raise RuntimeError("the modular reduction runs on the GPU only (no CPU fallback); "
                           "tests inject `local_modsum`")
    assert parts.dtype == torch.int64 and parts.is_contiguous()
    out = torch.empty(parts.shape[1], dtype=torch.int64, device=parts.device)
    stream = torch.cuda.current_stream(parts.device).cuda_stream
    capi.check(capi.load().sda_modsum_parts_dev(modulus, parts.data_ptr(), parts.shape[0], parts.shape[1],
                                                parts.shape[1], out.data_ptr(), stream or None))
    return out


def modular_allreduce(partial: torch.Tensor, modulus: int, group=None,
                      local_modsum: Optional[Callable[[torch.Tensor, int], torch.Tensor]] = None,
                      force_collectives: bool = False) -> torch.Tensor:
    """Sum of every rank's `partial` (int64 residues, any shape, same on all ranks) modulo `modulus`,
    returned on every rank.  `local_modsum(parts[G][len], modulus) -> [len]` defaults to the HIP kernel.
    `force_collectives`: a ONE-rank group still goes through all_to_all / all_gather (tests of the exchange on one
    device); an argument, not an environment variable - nothing in this package reads the environment to steer the path."""
    reduce_fn = local_modsum or _hip_modsum_parts
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    flat = partial.reshape(-1).contiguous()
    force = dist.is_initialized() and force_collectives                                 # exercise the exchange with 1 rank
    if world == 1 and not force:
        return reduce_fn(flat.unsqueeze(0), modulus).reshape(partial.shape)
    n = flat.numel()
    seg = (n + world - 1) // world                      # slice owned by each rank
    # gloo moves host memory only: with device tensors (a one-GPU rehearsal of the multi-rank path) the two exchanges are
    # staged through the host, the modular sum of the slices still runs on the device
    staged = flat.is_cuda and dist.get_backend(group) == "gloo"
    wire_dev = torch.device("cpu") if staged else flat.device
    padded = torch.zeros(seg * world, dtype=torch.int64, device=wire_dev)
    padded[:n] = flat.to(wire_dev)
    recv = torch.empty_like(padded)                     # [world][seg]: slice `rank` of every peer
    dist.all_to_all_single(recv, padded, group=group)
    mine = reduce_fn(recv.to(flat.device).view(world, seg), modulus)    # exact modular sum of the G slices
    gathered = torch.empty(seg * world, dtype=torch.int64, device=wire_dev)
    dist.all_gather_into_tensor(gathered, mine.contiguous().to(wire_dev), group=group)
    return gathered[:n].to(flat.device).reshape(partial.shape)
