"""Row-block sharding of the encrypted database across the GPUs of one node, and the result gather.

The reference processes the 16384-vector blocks of the database in a plain serial loop
(/root/reference/src/sender/sender_diag.cpp:28-30); blocks are independent, so rank r owns a contiguous range of
blocks, runs its own mat-vec, and the only exchange is the gather of result ciphertexts to rank 0 (RCCL over xGMI:
backend "nccl" on ROCm; "gloo" in the CPU tests).  torch is used for the collective only.
"""


def shard_blocks(total_blocks, world, rank):
    """Contiguous block range [lo, hi) of `rank`: the first total_blocks % world ranks take one extra block."""
    base, extra = divmod(total_blocks, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_vectors(n_total, slots, world, rank):
    """Vector range [lo, hi) of `rank` when the database of n_total vectors is cut on block (slots) boundaries."""
    G = -(-n_total // slots)
    blo, bhi = shard_blocks(G, world, rank)
    return min(blo * slots, n_total), min(bhi * slots, n_total)


def global_indices(local_indices, rank_first_vector):
    """decryptIndex returns j + i*batchSize inside a rank's shard (receiver_hers.cpp:46-49); shift to DB coordinates."""
    return [int(i) + int(rank_first_vector) for i in local_indices]


def gather_results(local, dist, rank, world, dst=0):
    """Gather equally-shaped result tensors (torch, on the collective's device) to `dst`; returns the list there."""
    import torch
    if world == 1:
        return [local]
    out = [torch.empty_like(local) for _ in range(world)] if rank == dst else None
    dist.gather(local, out, dst=dst)
    return out


def allreduce_membership_residues(local, moduli_per_row, dist):
    """Multi-GPU membership tail: residues are < 2^60, so an 8-way integer sum cannot overflow int64; all-reduce SUM
    then reduce each limb row modulo its prime (SURVEY.md §5 'Distributed communication backend').
    local: int64 tensor [polys*limbs, N]; moduli_per_row: python ints, one per row."""
    import torch
    dist.all_reduce(local, op=dist.ReduceOp.SUM)
    for r, q in enumerate(moduli_per_row):
        local[r] = torch.remainder(local[r], int(q))
    return local
