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
