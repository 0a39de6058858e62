"""World-size-2 `gloo` test of the sample-parallel path (ddmi_amd/dist.py): sharding, per-sample seeds, all-gather and
re-ordering give the same sample set as a single process."""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def fake_generate(indices, base_seed=777):
    """Stands in for ImagePipeline.generate on CPU: each sample is a pure function of (base_seed, global index)."""
    from ddmi_amd.dist import sample_seed
    rows = []
    for i in indices:
        g = torch.Generator().manual_seed(sample_seed(base_seed, i))
        rows.append(torch.randn((3, 4, 4), generator=g))
    return torch.stack(rows)


def _worker(rank, world, port, total, b_local, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from ddmi_amd import dist as D
    r, _, w = D.init_process_group("gloo")
    assert (r, w) == (rank, world)
    full = D.generate_sharded(fake_generate, total, b_local, r, w)
    t = D.max_over_ranks(float(rank + 1), "cpu")
    assert t == float(world)
    D.barrier()
    torch.save(full, os.path.join(out_dir, f"rank{rank}.pt"))
    torch.distributed.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_sharded_generation_matches_single_process(tmp_path):
    from ddmi_amd import dist as D
    total, b_local, world = 12, 3, 2
    mp.spawn(_worker, args=(world, _free_port(), total, b_local, str(tmp_path)), nprocs=world, join=True)
    ref = fake_generate(list(range(total)))
    for rank in range(world):
        got = torch.load(os.path.join(tmp_path, f"rank{rank}.pt"))
        assert torch.equal(got, ref), rank  # every rank ends with the full, index-ordered set
    # single-process path uses the same code
    assert torch.equal(D.generate_sharded(fake_generate, total, b_local, 0, 1), ref)


def test_self_launch_starts_the_ranks(tmp_path):
    """`python bench.py --gpus N` without torchrun: ddmi_amd.dist.launch_ranks starts N child ranks (here 2, gloo)."""
    import json
    from ddmi_amd import dist as D
    rc = D.launch_ranks(2, os.path.join(ROOT, "tests", "_rank_script.py"), [str(tmp_path)])
    assert rc == 0
    for rank in range(2):
        with open(os.path.join(tmp_path, f"rank{rank}.json")) as f:
            rec = json.load(f)
        assert rec["world"] == 2 and rec["ranks"] == 2 and rec["rows"] == [0.0, 0.0, 1.0, 1.0] and rec["max"] == 1.0
        # one set of convolution plans for the whole job: every rank finalizes from rank 0's file
        assert rec["finalized_from"] == "v 2\nplans tuned by rank 0\n"
        # a later finalize in the same process shares again (the first call's plan file is not mistaken for a user's) ...
        assert rec["second_from"] == "v 2\nplans tuned by rank 0\nsecond network, tuned by rank 0\n"
        # ... without leaving its pid-named plan path in the environment for child processes to inherit (round-3 advisor item)
        assert rec["env_after_finalize"] is None
        # ... and rank 0's failure is every rank's failure, not a hang in the broadcast
        assert rec["failure"] and "no such weight" in rec["failure"] and "finalized_after_failure" not in rec
        # the training step's bucketed gradient all-reduce: rank r held (r + 1) * arange(10) -> the sum is 3 * arange(10)
        assert rec["grad_slab"] == [3.0 * i for i in range(10)]
    # a failing rank is the launcher's failure
    assert D.launch_ranks(2, os.path.join(ROOT, "tests", "_rank_script.py"), [os.path.join(str(tmp_path), "missing")]) != 0


def test_shard_helpers():
    from ddmi_amd import dist as D
    assert D.shard_indices(10, 1, 4) == [1, 5, 9]
    assert sorted(sum((D.shard_indices(16, r, 8) for r in range(8)), [])) == list(range(16))
    assert D.sample_seed(777, 5) == 777 * 2 ** 32 + 5
    g = torch.arange(12).reshape(12, 1)            # rank-major gather of world=3, b_local=4: row r*4+j = index j*3+r
    idx = torch.tensor([[j * 3 + r] for r in range(3) for j in range(4)])
    assert torch.equal(D.reorder_rank_major(idx, 3, 4), g)


def test_config_loader_reads_reference_schema():
    from ddmi_amd.config import load_ldm_args
    a = load_ldm_args(os.path.join(ROOT, "configs", "ldm", "afhq.yaml"))
    assert a.unetconfig.model_channels == 224 and a.ddpmconfig.timesteps == a.ddpmconfig.sampling_timesteps == 250
    assert a.data_config.test_resolution == 256 and a.mode == "gen" and a.domain == "image"
    from ddmi_amd.ddpm import DDPM
    assert not DDPM(model=None, **a.ddpmconfig).is_ddim_sampling          # -> p_sample_loop
    b = load_ldm_args(os.path.join(ROOT, "configs", "ldm", "afhq_ddim50.yaml"))
    assert DDPM(model=None, **b.ddpmconfig).is_ddim_sampling


def _overlap_worker(rank, world, port, out_dir, hooks_after_late=0):
    """A fake backward on two gloo ranks: the slab's ranges complete from its end, buckets go out while later ranges are still
    'computed', the late views are written last - against the blocking all-reduce of the same gradients."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from ddmi_amd import dist as D
    from ddmi_amd.train import OverlappedGradReducer, allreduce_gradients
    D.init_process_group("gloo")
    g = torch.Generator().manual_seed(1234 + rank)
    n = 1000
    local = torch.randn(n, generator=g)                       # this rank's gradients
    # slab layout: [head 0:100 | b1 100:300 | b2 300:450 | b3 450:800 | b4 800:990 | logit 990:1000]; the backward completes logit, b4 .. b1
    ranges = [(990, 1000), (800, 990), (450, 800), (300, 450), (100, 300)]
    late = [(120, 140), (310, 330), (500, 540), (900, 910), (20, 30)]       # "emb_layers" views (one of them in the head)
    blocking = local.clone()
    allreduce_gradients(blocking, world, bucket_bytes=4 * 128)
    slab = torch.zeros(n)
    red = OverlappedGradReducer(slab, ranges, late, world, bucket_bytes=4 * 256)

    def write(lo, hi):                                          # a block's backward: everything but the late views
        slab[lo:hi] = local[lo:hi]
        for a, b in late:
            a2, b2 = max(a, lo), min(b, hi)
            if a2 < b2:
                slab[a2:b2] = 0.0
    n_before = len(ranges) - hooks_after_late
    for i, (lo, hi) in enumerate(ranges[:n_before]):
        write(lo, hi)
        red.range_done(i)
    for lo, hi in ranges[n_before:]:
        write(lo, hi)
    write(0, 100)                                               # the head (input_blocks.0, time_embed): no hook, left to finish()
    red.before_late_writes()
    for a, b in late:                                           # EmbProjFn.backward: the late views, accumulated into the slab
        slab[a:b] += local[a:b]
    # hooks that fire AFTER the late writes (nothing but autograd's scheduling keeps a block-input hook in front of EmbProjFn.backward):
    # a bucket issued now would carry final late values that finish() then reduces a second time - range_done() must only record
    for i in range(n_before, len(ranges)):
        red.range_done(i)
    red.finish()
    torch.save({"overlapped": slab, "blocking": blocking, "collectives": red.collectives}, os.path.join(out_dir, f"ov{rank}.pt"))
    D.barrier()
    torch.distributed.destroy_process_group()


def test_overlapped_gradient_allreduce_equals_the_blocking_one(tmp_path):
    """tools/ldm/image.py:96, 165 (DDP's bucketed reduction during the backward): OverlappedGradReducer on two gloo ranks gives the
    blocking all-reduce's slab BIT FOR BIT, with the buckets issued before the 'backward' has finished."""
    world = 2
    mp.spawn(_overlap_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for rank in range(world):
        rec = torch.load(os.path.join(tmp_path, f"ov{rank}.pt"))
        assert torch.equal(rec["overlapped"], rec["blocking"]), rank
        assert rec["collectives"] >= 3          # buckets during the backward + the head + the gathered late views


def test_overlapped_allreduce_with_hooks_firing_after_the_late_writes(tmp_path):
    """Round-4 advisor item: a range_done() after before_late_writes() must not send a bucket (its late views are final by then and
    finish() would sum them over the ranks twice).  Two of the five block hooks fire late here; the slab still equals the blocking pass."""
    world = 2
    mp.spawn(_overlap_worker, args=(world, _free_port(), str(tmp_path), 2), nprocs=world, join=True)
    for rank in range(world):
        rec = torch.load(os.path.join(tmp_path, f"ov{rank}.pt"))
        assert torch.equal(rec["overlapped"], rec["blocking"]), rank


def test_reduce_plan_ranges_follow_the_slab_layout():
    """The completion-ordered ranges OverlappedGradReducer takes are contiguous and descending for any block structure (checked on the
    helper's own invariants with a hand-made layout; the real plan is asserted on the GPU by the trainer's constructor)."""
    from ddmi_amd.train import OverlappedGradReducer
    slab = torch.zeros(100)
    OverlappedGradReducer(slab, [(90, 100), (40, 90), (10, 40)], [(50, 60)], 1)
    import pytest
    with pytest.raises(AssertionError):
        OverlappedGradReducer(slab, [(90, 100), (30, 80)], [], 1)      # a gap: not the tail of the slab


def _bench_dry_line(argv, env=None):
    import json
    import subprocess
    r = subprocess.run([sys.executable] + argv, cwd=ROOT, capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout   # ONE JSON line, from rank 0 only
    return json.loads(lines[0])


def test_bench_dry_run_two_ranks_self_launched():
    """Round-5 verdict item 4: bench.py's own rank plumbing (launch_ranks -> torchrun env -> init_process_group -> shared-plan
    broadcast -> all_gather_samples -> max_over_ranks -> one JSON line) had never executed anywhere.  `python bench.py --gpus 2
    --dry-run` runs exactly that on CPU tensors over gloo with a stub step."""
    line = _bench_dry_line(["bench.py", "--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1"])
    assert line["dry_run"] is True and line["backend"] == "gloo"
    assert line["n_gpus"] == 2 and line["nccl_ranks"] == 2 and line["steps"] == 3 and line["warmup"] == 1
    assert line["config"]["global_batch"] == 2 * line["config"]["batch_per_gpu"]
    assert line["config"]["conv_plans"].startswith("shared")
    assert line["scaling"] == "weak" and line["value"] > 0 and line["ms_per_step"] > 0


def test_bench_dry_run_under_the_drivers_torchrun_command():
    """The driver's form for N > 1: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...` - the ranks read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment."""
    line = _bench_dry_line(["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                            "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run"])
    assert line["n_gpus"] == 2 and line["nccl_ranks"] == 2 and line["config"]["global_batch"] == 8


def test_bench_dry_run_single_process():
    line = _bench_dry_line(["bench.py", "--dry-run"])
    assert line["n_gpus"] == 1 and line["nccl_ranks"] == 1 and line["dry_run"] is True
