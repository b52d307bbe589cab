"""image_matching_amd — MI355X-native HyDia (approach 5) encrypted similarity search.

The product is libhydia.so (hand-written gfx950 HIP kernels + C++ host, C-ABI in include/hydia.h).  This package is
the thin Python host mirror of the reference's role classes over that C-ABI; it never computes on the CPU and
raises if the HIP library is missing.
"""
import os as _os

# two comparator lanes + whatever streams the host application (torch, RCCL) creates: more than ROCm's default 4 hardware
# queues, or the lanes share a queue and serialise.  Only effective if set before the HIP runtime initialises.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from .hydia import (Context, Ciphertext, DiagonalEnroller, DiagonalReceiver, DiagonalSender, HydiaError,  # noqa: F401
                    HersEnroller, HersReceiver, HersSender,
                    byte_ledger, default_params, describe_params, compute_required_depth, lib_path, load_library)
from .sharding import (ShardGroup, ShardedDiagonalEnroller, ShardedDiagonalSender, DistDiagonalEnroller,  # noqa: F401
                       DistDiagonalSender, group_babies, shard_blocks, shard_vectors)

MATCH_THRESHOLD = 0.44  # include/config.h:9
COMP_DEPTH = 10         # include/config.h:14
VECTOR_DIM = 512        # include/config.h:30
