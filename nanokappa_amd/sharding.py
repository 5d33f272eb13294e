"""Particle sharding rules shared by the host code and documented for the kernels.

The ensemble is partitioned by particle; every rank holds all tables.  There is no data-path exchange of particles:
the only collective is the per-step sum of the tally vector (RCCL all-reduce inside nk_step).
"""


def shard_range(n, rank, nranks):
    """Initial particles [lo, hi) owned by `rank` (even split by index)."""
    return (n * rank) // nranks, (n * (rank + 1)) // nranks


def emission_owner(rm, level, step, nranks):
    """Rank that creates the `level`-th particle of reservoir-mode entry `rm` at `step`.  Every rank advances all
    reservoir counters identically and keeps only its own entries (k_emit_count in csrc/nk_kernels.h)."""
    return (rm + level + step) % nranks


def emission_pid(rm, level, step):
    """64-bit id of an emitted particle: (step+1) mod 2^24 | rm (28 bits) | level (12 bits).  Ids key the counter-based
    RNG, so a particle's random decisions do not depend on which rank or slot holds it."""
    return (((step + 1) & 0xFFFFFF) << 40) | (rm << 12) | level
