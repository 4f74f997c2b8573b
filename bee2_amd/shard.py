"""Index arithmetic for sharding a batch of independent items over G GPUs.
No data-path collective exists: rank g owns [lo, hi) and, for CTR, starts its counter at
block offset lo (bee2hip_beltCTR_blocks_dev's first_block)."""


def shard_range(rank, world, n):
    """contiguous, balanced: ranks differ by at most one item"""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    lo = n * rank // world
    hi = n * (rank + 1) // world
    return lo, hi


def ctr_shard(rank, world, nbytes):
    """(byte_lo, byte_hi, first_block) for a stream of nbytes: whole 16-byte blocks per rank,
    the last rank also takes the ragged tail"""
    nblocks = nbytes // 16
    lo, hi = shard_range(rank, world, nblocks)
    byte_hi = hi * 16 if rank < world - 1 else nbytes
    return lo * 16, byte_hi, lo


def broadcast_params(dist, blob, device=None):
    """rank 0's parameter block (expanded key || ctr0, or curve constants) to every rank;
    works over RCCL ('nccl') on GPUs and over gloo on CPU"""
    import torch
    t = torch.zeros(len(blob), dtype=torch.uint8, device=device)
    if dist.get_rank() == 0:
        t.copy_(torch.frombuffer(bytearray(blob), dtype=torch.uint8))
    dist.broadcast(t, src=0)
    return t.cpu().numpy().tobytes()
