"""Image-level sharding for multi-GPU inference (one process per GPU).

The reference has no multi-GPU code; BASELINE.json asks for "simple image-sharded data parallel for
inference".  The path shards by image (independent units, val.py:157-158 processes one image per step),
so there is NO collective in the data path: rank r takes images r, r+W, r+2W, ... and only the per-image
scalar results are gathered once at the end (python objects over the process group; gloo or nccl/RCCL).
"""
import torch.distributed as dist


def rank_and_world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def barrier(group=None):
    """all ranks meet (no-op without a process group)"""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.barrier(group=group)


def shard_indices(n_items, rank, world):
    """round-robin shard: indices rank, rank + world, ...  (every item exactly once over all ranks)."""
    if not (0 <= rank < world):
        raise ValueError('rank {} outside world {}'.format(rank, world))
    return list(range(rank, n_items, world))


def shard(items, rank=None, world=None):
    if rank is None or world is None:
        rank, world = rank_and_world()
    return [items[i] for i in shard_indices(len(items), rank, world)]


def gather_in_order(local_results, n_items, group=None):
    """local_results: list of (global_index, value) computed by this rank.  Returns on EVERY rank the list of
    values ordered by global index.  Raises if an index is missing or duplicated."""
    rank, world = rank_and_world(group)
    if world == 1:
        parts = [list(local_results)]
    else:
        parts = [None] * world
        dist.all_gather_object(parts, list(local_results), group=group)
    out = [None] * n_items
    seen = 0
    for part in parts:
        for idx, val in part:
            if out[idx] is not None:
                raise RuntimeError('item {} produced twice'.format(idx))
            out[idx] = val
            seen += 1
    if seen != n_items:
        raise RuntimeError('expected {} results, gathered {}'.format(n_items, seen))
    return out
