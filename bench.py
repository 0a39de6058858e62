#!/usr/bin/env python3
"""Headline benchmark: samples/sec at 256x256, 250-step DDPM + INR decode (BASELINE.json), one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A *step* is one pass of the hot path over one batch: x_T -> 250 x (UNet + fused DDPM update, ancestral
p_sample_loop with timesteps=250) -> D2C-VAE decode -> fused plane sampling + coordinate MLP -> clamp, for B samples
per GPU on random-init (named-seed) AFHQ-architecture weights, bf16 operands / fp32 accumulate, inputs and noise
generated on the device.  With N GPUs every rank runs independent chains (weak scaling) and the finished images are
all-gathered over RCCL inside the timed region.  Rank 0 prints ONE JSON line.

Besides the contract fields the line carries
  roofline     - the dominant kernel (the MFMA implicit-GEMM convolution) timed live with HIP events on its launch
                 stream during the timed region (every 25th UNet forward), algorithmic FLOPs / time vs 2.5 PFLOP/s
  cpu_baseline - the CPU oracle (`oracle/ddmi_oracle.py`, a port of the reference's PyTorch-CPU path) timed on this
                 box's host cores on a bounded sample: ONE full-size UNet evaluation at B=1 (1 of the 250 that make up
                 98.8 % of a sample's work), extrapolated to samples/sec.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

AFHQ = dict(
    unetconfig=dict(image_size=64, in_channels=64, model_channels=224, out_channels=64, attention_resolutions=[8, 4, 2],
                    num_res_blocks=2, channel_mult=[1, 2, 4, 6], num_head_channels=32),
    ddconfig=dict(double_z=True, z_channels=128, resolution=256, in_channels=3, out_ch=64, ch=128, ch_mult=[1, 2, 4],
                  num_res_blocks=2, attn_resolutions=[], hdbf_resolutions=[128, 64], dropout=0.0, attn_type="vanilla"),
    mlpconfig=dict(in_ch=2, out_ch=3, ch=256, latent_dim=64),
    embed_dim=64,
)
NFE = 250
PEAK_BF16_TFLOPS = 2500.0  # dense, /opt/skills/guides/MI355X_MICROARCH.md


def cpu_baseline(threads=None):
    """Times the oracle's UNet on the host (kind 'port'): one evaluation at B=1, full AFHQ size."""
    import torch
    from ddmi_amd import initweights as iw
    from ddmi_amd.unet import unet_manifest
    from oracle import ddmi_oracle as O
    # physical cores: SMT siblings only oversubscribe the fp32 GEMMs (measured: 256 threads are 3x slower than 128
    # on the 2 x 64-core host of the GPU box)
    cores = threads or max(1, (os.cpu_count() or 2) // 2)
    torch.set_num_threads(cores)
    cfg = AFHQ["unetconfig"]
    st = O.to_state(iw.fill_state("afhq.unet", unet_manifest(**cfg)))
    x = torch.from_numpy(iw.normal("bench", "x", (1, 64, 64, 64)))
    t = torch.tensor([125], dtype=torch.long)
    with torch.no_grad():
        t0 = time.perf_counter()
        O.unet_forward(st, cfg, x, t)
        dt = time.perf_counter() - t0
    return {"value": 1.0 / (NFE * dt), "unit": "samples/sec", "cores": cores, "kind": "port",
            "sample": f"1 of {NFE} UNet evaluations at B=1, 64x64x64 latent ({dt:.2f} s), x{NFE}; decode+MLP (1.2 % of the work) not included",
            "unet_eval_s": dt}


def pmc_traffic(live_name):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/*_pmc_traffic.json:
    separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of this script, gfx950 corrections applied)."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic*.json")))
    if not files or not live_name:
        return None
    m = re.match(r"conv_ring<bf16,(\d+)x(\d+),k(\d+)x(\d+)>", live_name)
    m2 = re.match(r"conv_(pipe|halo)<bf16,(\d+)x(\d+)>", live_name)
    if m:
        bm, bn, kk, nst = (int(v) for v in m.groups())
        pat = re.compile(rf"conv_ring_kernel<{bm}, {bn}, \d+, \d+, {kk // 32}, {nst}>")
    elif m2 and m2.group(1) == "halo":  # conv_halo_kernel<BN, WM, WN>: BM is fixed at 256
        pat = re.compile(rf"conv_halo_kernel<{m2.group(3)}, ")
    elif m2:
        pat = re.compile(rf"conv_pipe_kernel<{m2.group(2)}, {m2.group(3)}, ")
    else:
        return None
    with open(files[-1]) as f:
        table = json.load(f)["kernels"]
    for name, v in table.items():
        if pat.search(name):
            return v["hbm_bytes_per_launch"]
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=512, help="samples per GPU per step")
    ap.add_argument("--nfe", type=int, default=NFE, help="diagnostics only; the headline metric is 250")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--seed", type=int, default=777)
    ap.add_argument("--rng", default="philox", choices=["philox", "torch"],
                    help="per-chain noise: in-kernel Philox streams (default) or one torch generator per chain")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher.  Nothing in this process has touched the GPU yet (no
        # HIP call), the ranks are CHILD processes (never an exec of this one) and their status is ours.
        from ddmi_amd.dist import launch_ranks
        sys.exit(launch_ranks(args.gpus, os.path.abspath(__file__), sys.argv[1:]))

    import torch
    from ddmi_amd import _lib
    from ddmi_amd import dist as D
    from ddmi_amd.pipeline import ImagePipeline

    rank, local_rank, world = D.init_process_group()
    assert world == args.gpus or world == 1 and args.gpus == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    dev = f"cuda:{local_rank}"
    torch.cuda.set_device(local_rank)
    B = args.batch

    ddpmconfig = dict(timesteps=args.nfe, image_size=64, channels=64, clip_denoised=False, mixed_prediction=True,
                      mixed_init=-6, sampling_timesteps=args.nfe, ddim_sampling_eta=1.0)  # S == T -> p_sample_loop
    pipe = ImagePipeline(AFHQ["unetconfig"], AFHQ["ddconfig"], AFHQ["mlpconfig"], ddpmconfig, AFHQ["embed_dim"],
                         test_resolution=256, numerics="bf16", max_batch=B, device=dev)
    pipe.load_named_seed("afhq").finalize()

    # primer (not a step): two denoiser calls + decode + MLP so lazy one-time setup is out of the timed region
    z0 = torch.randn((B, 64, 64, 64), device=dev)
    tt = torch.full((B,), 3, device=dev, dtype=torch.int64)
    for _ in range(2):
        pipe.unet(z0, tt)
    pipe.mlp(pipe.coords(), hdbf=pipe.vae.decode(z0), si=1.0)
    torch.cuda.synchronize()

    phase = {"sample": 0.0, "decode_mlp": 0.0, "gather": 0.0}

    def one_step(step_idx):
        idx = [(step_idx * world + rank) * B + j for j in range(B)]  # disjoint global sample indices
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        ev[0].record()
        img = pipe.generate(idx, base_seed=args.seed, events=ev[1:3], rng=args.rng)
        out = D.all_gather_samples(img, world)
        ev[3].record()
        phase["_ev"] = ev
        return out

    for w in range(args.warmup):
        one_step(w)
    _lib.profile_enable(25)
    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = None
    for k in range(args.steps):
        out = one_step(args.warmup + k)
    torch.cuda.synchronize()
    D.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = D.max_over_ranks(elapsed, dev)
    ev = phase.pop("_ev")
    phase_ms = {"sampling_loop": ev[0].elapsed_time(ev[1]), "decode": ev[1].elapsed_time(ev[2]),
                "mlp_clamp_gather": ev[2].elapsed_time(ev[3])}
    prof = _lib.profile_collect()
    _lib.profile_enable(0)
    assert out is not None and out.shape == (world * B, 3, 256, 256) and bool(torch.isfinite(out).all())

    if rank == 0:
        total_samples = args.steps * world * B
        conv = {k: v for k, v in prof.items() if k.startswith("conv_")}
        dom = max(conv, key=lambda k: conv[k]["total_ms"]) if conv else None
        roof = None
        if dom:
            d = conv[dom]
            ach = d["flops"] / (d["total_ms"] * 1e-3) / 1e12
            roof = {"bound": "mfma", "kernel": dom, "achieved": round(ach, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": pmc_traffic(dom),
                    "algorithmic_bytes_per_launch": round(d["bytes"] / d["launches"]), "launches_timed": d["launches"],
                    "avg_launch_us": round(1e3 * d["total_ms"] / d["launches"], 2),
                    "avg_launch_gflop": round(d["flops"] / d["launches"] / 1e9, 3)}
        tot_ms = sum(v["total_ms"] for v in prof.values()) or 1.0
        breakdown = {k: round(v["total_ms"] / tot_ms, 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])}
        line = {
            "metric": "samples/sec (whole node) at 256x256 image, 250-step DDPM + INR decode",
            "value": total_samples / elapsed, "unit": "samples/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"AFHQ 256x256, {args.nfe}-step DDPM (p_sample_loop, T={args.nfe}) + D2C-VAE decode + "
                                   "fused HDPE plane sampling/coordinate MLP, random-init named-seed weights",
                       "batch_per_gpu": B, "global_batch": world * B, "nfe": args.nfe, "latent": [64, 64, 64],
                       "parallelism": f"sample-parallel x{world} (independent chains, all-gather of finished images)",
                       "noise": "per-chain Philox streams drawn in the sampler-step kernel" if args.rng == "philox"
                                else "one torch device generator per chain"},
            "roofline": roof, "kernel_time_share": breakdown,
            "last_step_phase_ms": {k: round(v, 2) for k, v in phase_ms.items()},
            "unet_memory_bytes": pipe.unet.memory(),
        }
        if not args.no_cpu_baseline and world == 1:  # a reported baseline: rank 0 at N = 1 only
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
