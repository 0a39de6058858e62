"""Sample-parallel generation over the GPUs of one node.

Generation is embarrassingly parallel per sample (independent Markov chains; GroupNorm and attention are per-sample),
so there is NO collective on the data path: every rank holds a full weight replica, owns the global sample indices
`i = rank, rank + world, ...`, seeds each chain from (base_seed, i) - so the set of samples does not depend on the
number of GPUs - and finished images are all-gathered once per batch (RCCL over xGMI with the `nccl` backend; `gloo`
in the CPU tests).  The reference never shards sampling: it builds the EMA model on the main process only
(tools/ldm/image.py:63-65), which is exactly what is NOT reproduced here.
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def env_rank_world() -> Tuple[int, int, int]:
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_process_group(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Initialises torch.distributed from the torchrun environment (no-op for a single process)."""
    rank, local_rank, world = env_rank_world()
    # DDMI_FORCE_GROUP=1: also form the (one-rank) group for WORLD_SIZE=1 - lets a 1-GPU box exercise the RCCL calls
    if (world > 1 or os.environ.get("DDMI_FORCE_GROUP") == "1") and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def launch_ranks(n: int, script: str, argv: Sequence[str], extra_env: Optional[dict] = None) -> int:
    """Starts `script argv...` as n ranks of one node (`python -m torch.distributed.run`, rendezvous on 127.0.0.1 at a
    free port) and returns the launcher's exit status.  For entry points invoked WITHOUT torchrun (`python bench.py
    --gpus 8`): call this before anything initialises the GPU - the ranks are child processes, never an exec of a
    process that holds a HIP context."""
    import socket
    import subprocess
    import sys
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(n)}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), script] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this driver (RCCL needs it)
    env.update(extra_env or {})
    return subprocess.call(cmd, env=env)


def group_size(device) -> int:
    """Number of ranks that take part in a collective right now (sum of ones over the group): what bench.py reports as
    `nccl_ranks`."""
    if not dist.is_initialized():
        return 1
    t = torch.ones(1, dtype=torch.int32, device=device)
    dist.all_reduce(t)
    return int(t.item())


_SHARED_PLAN_PATH: Optional[str] = None   # this process's plan file for shared finalizes (a module variable, NOT an inherited env var)


def _drop_shared_plan_file() -> None:
    global _SHARED_PLAN_PATH
    if _SHARED_PLAN_PATH is not None:
        try:
            os.unlink(_SHARED_PLAN_PATH)
        except OSError:
            pass
        _SHARED_PLAN_PATH = None


def finalize_with_shared_plans(finalize: Callable[[], object], rank: int, world: int) -> Optional[str]:
    """Runs `finalize()` (pipeline.finalize: weight packing + per-shape convolution autotune) on every rank with ONE set of
    convolution plans: rank 0 tunes into a plan file, its text is broadcast, the other ranks finalize from it.  The
    autotuner is timing-based, so two ranks tuning on their own can pick different tiles / split-K for the same shape;
    with shared plans the bf16 results of a sample are bit-identical whichever GPU produced it.  A user-pinned
    DDMI_TUNE_FILE is respected as is.  The per-process plan file this function creates is remembered in a MODULE variable and
    put into the environment only while `finalize()` runs (the library reads DDMI_TUNE_FILE at finalize): a LATER call in the
    same process runs the same protocol again on the same file (rank 0 tunes the new network's shapes into it, the others receive
    the whole text), and a child process started in between (tests, the self-launcher, a build) never inherits a pid-named path it
    would overwrite or outlive.  The file is removed at exit.  If rank 0's finalize raises, the error text is broadcast and every
    rank raises - nobody waits in the collective.  Returns the plan file path."""
    global _SHARED_PLAN_PATH
    if world == 1 or not dist.is_initialized():
        finalize()
        return os.environ.get("DDMI_TUNE_FILE")
    if os.environ.get("DDMI_TUNE_FILE"):   # the user's file: never ours (ours is not left in the environment)
        finalize()
        return os.environ.get("DDMI_TUNE_FILE")
    if _SHARED_PLAN_PATH is None:
        import atexit
        import tempfile
        _SHARED_PLAN_PATH = os.path.join(tempfile.gettempdir(), f"ddmi_conv_plans_{os.getpid()}_rank{rank}.txt")
        atexit.register(_drop_shared_plan_file)
    path = _SHARED_PLAN_PATH

    def finalize_on_shared_file():
        os.environ["DDMI_TUNE_FILE"] = path
        try:
            finalize()
        finally:
            os.environ.pop("DDMI_TUNE_FILE", None)

    msg = [None, None]   # (plan text, error text)
    if rank == 0:
        try:
            finalize_on_shared_file()
            msg[0] = open(path).read() if os.path.exists(path) else ""
        except BaseException as e:   # the other ranks are waiting in the broadcast: tell them before re-raising
            msg[1] = f"{type(e).__name__}: {e}"
            dist.broadcast_object_list(msg, src=0)
            raise
    dist.broadcast_object_list(msg, src=0)
    if msg[1] is not None:
        raise RuntimeError(f"rank 0 failed to finalize: {msg[1]}")
    if rank != 0:
        with open(path, "w") as f:
            f.write(msg[0])
        finalize_on_shared_file()
    return path


def shard_indices(total: int, rank: int, world: int) -> List[int]:
    """Global sample indices owned by `rank` (round-robin, so any prefix of the index space is balanced)."""
    return list(range(rank, total, world))


def sample_seed(base_seed: int, index: int) -> int:
    """Per-chain seed: base_seed * 2^32 + global sample index."""
    return (int(base_seed) << 32) + int(index)


def barrier() -> None:
    if dist.is_initialized():
        dist.barrier()


def all_gather_samples(local: torch.Tensor, world: int) -> torch.Tensor:
    """[B_local, ...] on every rank -> [world * B_local, ...] in rank-major order (one all-gather per batch)."""
    if not dist.is_initialized():
        return local
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous())
    return out


def reorder_rank_major(gathered: torch.Tensor, world: int, b_local: int) -> torch.Tensor:
    """Undo the round-robin sharding: row r*b_local + j of the gather is global index j*world + r."""
    if world == 1:
        return gathered
    g = gathered.reshape(world, b_local, *gathered.shape[1:])
    return g.transpose(0, 1).reshape(world * b_local, *gathered.shape[1:])


def max_over_ranks(value: float, device) -> float:
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def generate_sharded(generate_fn: Callable[[Sequence[int]], torch.Tensor], total: int, b_local: int, rank: int,
                     world: int) -> torch.Tensor:
    """Runs `generate_fn(indices) -> [len(indices), ...]` over this rank's shard in batches of b_local and returns the
    full, index-ordered sample set on every rank.  `total` must be a multiple of world * b_local."""
    assert total % (world * b_local) == 0, "total must be a multiple of world_size * per-GPU batch"
    mine = shard_indices(total, rank, world)
    chunks = []
    for k in range(0, len(mine), b_local):
        local = generate_fn(mine[k:k + b_local])
        chunks.append(reorder_rank_major(all_gather_samples(local, world), world, b_local))
    return torch.cat(chunks, dim=0)
