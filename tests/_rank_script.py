"""Run by tests/test_dist_gloo.py through ddmi_amd.dist.launch_ranks (the `python bench.py --gpus N` self-launch path):
one rank of a world the launcher started; writes what it saw to argv[1]."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from ddmi_amd import dist as D  # noqa: E402

rank, local_rank, world = D.init_process_group()  # gloo on a CPU-only host, nccl (= RCCL) on a GPU box
dev = f"cuda:{local_rank}" if torch.cuda.is_available() else "cpu"
local = torch.full((2, 3), float(rank), device=dev)
full = D.all_gather_samples(local, world)
rec = {"rank": rank, "world": world, "ranks": D.group_size(dev), "rows": full[:, 0].tolist(),
       "backend": torch.distributed.get_backend() if torch.distributed.is_initialized() else None,
       "max": D.max_over_ranks(float(rank), dev)}


def fake_finalize():
    """What a finalize does with DDMI_TUNE_FILE (net.hip): reads the plan file if it exists, else tunes and writes it."""
    p = os.environ.get("DDMI_TUNE_FILE")
    if p and not os.path.exists(p):
        with open(p, "w") as f:
            f.write(f"v 2\nplans tuned by rank {rank}\n")
    rec["finalized_from"] = open(p).read() if p else None


# the training step's gradient all-reduce (ddmi_amd/train.py): bucketed sum of the flat slab, in place
from ddmi_amd.train import allreduce_gradients  # noqa: E402
slab = torch.arange(10, dtype=torch.float32, device=dev) * (rank + 1)
allreduce_gradients(slab, world, bucket_bytes=16)          # 4-element buckets: three collectives
rec["grad_slab"] = slab.tolist()
os.environ.pop("DDMI_TUNE_FILE", None)
D.finalize_with_shared_plans(fake_finalize, rank, world)
# the per-process plan file is remembered in the module, not left in the environment for child processes to inherit
rec["env_after_finalize"] = os.environ.get("DDMI_TUNE_FILE")


def second_finalize():
    """A second network finalized later in the same process: rank 0 appends the shapes it had to tune; the other ranks must see
    the appended text (not tune on their own because the first call left DDMI_TUNE_FILE set)."""
    p = os.environ.get("DDMI_TUNE_FILE")
    if p is None:      # a single process shares nothing: no plan file is forced on it
        rec["second_from"] = None
        return
    if rank == 0:
        with open(p, "a") as f:
            f.write("second network, tuned by rank 0\n")
    rec["second_from"] = open(p).read()


D.finalize_with_shared_plans(second_finalize, rank, world)


def failing_finalize():
    if rank == 0:
        raise ValueError("no such weight")
    rec["finalized_after_failure"] = True


try:   # rank 0's error reaches every rank instead of leaving them in the broadcast
    D.finalize_with_shared_plans(failing_finalize, rank, world)
    rec["failure"] = None
except (RuntimeError, ValueError) as e:
    rec["failure"] = str(e)
with open(os.path.join(sys.argv[1], f"rank{rank}.json"), "w") as f:
    json.dump(rec, f)
D.barrier()
if torch.distributed.is_initialized():
    torch.distributed.destroy_process_group()
