#!/usr/bin/env python3
"""Entry point with the reference's surface: `python main.py --exp ldm --configs configs/ldm/afhq.yaml [--seed 777]`
(reference main.py:54-65), also under `accelerate launch` / `torch.distributed.run` (one process per GPU).

Only the generation path is implemented (`data.mode: gen | eval` of the `ldm` experiment; image, video and occupancy domains;
occupancy writes one Wavefront .obj per sample, tools/ldm/occupancy.py:205-230): image gen writes a
grid of `test_batch_size` samples like tools/ldm/image.py:225-249; eval runs the sample-parallel 10k-image loop of
evals/eval.py:215-234 (images written as PNG shards; Inception scoring needs downloaded weights and is out of scope).
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main(args):
    if args.exp != "ldm":
        raise SystemExit("only --exp ldm (generation) is implemented; stage-1 training is out of scope (SURVEY.md 8)")
    import numpy as np
    import torch
    from ddmi_amd import dist as D
    from ddmi_amd.config import load_ldm_args
    from ddmi_amd.coords import random_seed
    from ddmi_amd.pipeline import ImagePipeline

    args = load_ldm_args(args.configs, args)
    if args.domain not in ("image", "video", "occupancy") or args.DiT:
        raise SystemExit(f"domain '{args.domain}' / DiT is out of scope (SURVEY.md 8): image, video and occupancy generation only")
    random_seed(args.seed)
    rank, local_rank, world = D.init_process_group()
    dev = f"cuda:{local_rank}"
    torch.cuda.set_device(local_rank)
    dc = args.data_config
    B = int(dc.test_batch_size)
    if args.domain == "occupancy":  # exp/stage.py:207-241, tools/ldm/occupancy.py:205-230
        return occupancy_main(args, dc, B, rank, world, dev)
    if args.domain == "video":  # exp/stage.py:245-262, tools/ldm/video.py:214-236
        from ddmi_amd.pipeline import VideoPipeline
        pipe = VideoPipeline(args.unetconfig, args.ddconfig, args.mlpconfig, args.ddpmconfig, args.embed_dim,
                             frames=int(dc.get("frames", 16)), test_resolution=int(dc.test_resolution),
                             numerics=args.numerics, max_batch=B, device=dev)
    else:
        pipe = ImagePipeline(args.unetconfig, args.ddconfig, args.mlpconfig, args.ddpmconfig, args.embed_dim,
                             test_resolution=int(dc.test_resolution), numerics=args.numerics, max_batch=B, device=dev)
    ckpt = os.path.join(dc.save_pth, "ldm-last.pt")
    if os.path.exists(ckpt):
        print("Loading Pretrained Models!")
        pipe.load_checkpoint(ckpt)
    else:
        print(f"{ckpt} not found: using named-seed synthetic weights")
        pipe.load_named_seed("sky" if args.domain == "video" else "afhq")
    pipe.finalize()
    out_dir = os.path.join(dc.save_pth, "results")
    os.makedirs(out_dir, exist_ok=True)
    if args.mode == "gen":
        print("Random image generation!")
        img = D.generate_sharded(lambda idx: pipe.generate(idx, base_seed=args.seed), world * B, B, rank, world)
        if rank == 0 and args.domain == "video":  # [N, 3, T, H, W] -> uint8 [N, T, H, W, 3] (tools/ldm/video.py:231-236 writes frames)
            u8 = ((img + 1) / 2 * 255).round().clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 4, 1).cpu().numpy()
            np.save(os.path.join(out_dir, "generation_video.npy"), u8)
            print("Finished generating videos!")
        elif rank == 0:
            u8 = ((img + 1) / 2 * 255).round().clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).cpu().numpy()
            np.save(os.path.join(out_dir, "generation.npy"), u8)
            try:
                from PIL import Image
                grid = np.concatenate(list(u8), axis=1)
                Image.fromarray(grid).save(os.path.join(out_dir, "generation.jpg"))
            except ImportError:
                pass
            print("Finished generating images!")
    elif args.mode == "eval":
        total = int(os.environ.get("DDMI_EVAL_SAMPLES", 10000)) // (world * B) * (world * B)
        img = D.generate_sharded(lambda idx: pipe.generate(idx, base_seed=args.seed), total, B, rank, world)
        if rank == 0:
            u8 = ((img + 1) / 2 * 255).round().clamp(0, 255).to(torch.uint8).cpu().numpy()
            np.save(os.path.join(out_dir, "fid_samples.npy"), u8)
            print(f"wrote {total} samples")
    else:
        raise SystemExit("data.mode must be gen or eval (training is out of scope)")


def generator_settings(dc):
    """What convocc's get_generator (conv_onet/config.py:117-128) reads: the convocc yaml named by data.conv_config over
    convocc/configs/default.yaml when that tree is present, else the `data.generation` block of our config."""
    import yaml
    g = dict(dc.get("generation") or {})
    out = {"threshold": float(g.get("threshold", 0.5)), "resolution0": int(g.get("resolution_0", 256)),
           "upsampling_steps": int(g.get("upsampling_steps", 0)), "padding": float(g.get("padding", 0.1))}
    for path in ("convocc/configs/default.yaml", dc.get("conv_config")):
        if path and os.path.exists(path):
            with open(path) as f:
                c = yaml.safe_load(f) or {}
            out["threshold"] = float(c.get("test", {}).get("threshold", out["threshold"]))
            out["resolution0"] = int(c.get("generation", {}).get("resolution_0", out["resolution0"]))
            out["upsampling_steps"] = int(c.get("generation", {}).get("upsampling_steps", out["upsampling_steps"]))
            out["padding"] = float(c.get("data", {}).get("padding", out["padding"]))
    return out


def occupancy_main(args, dc, B, rank, world, dev):
    """`gen`: one batch of meshes per rank -> results/generation-<index>.obj (tools/ldm/occupancy.py:222-230 writes one);
    `eval`: the 5 000-shape loop of tools/ldm/occupancy.py:205-219, sharded over the ranks by global sample index."""
    from ddmi_amd.pipeline import OccupancyPipeline
    pipe = OccupancyPipeline(args.unetconfig, args.ddconfig, args.mlpconfig, args.ddpmconfig, args.embed_dim,
                             numerics=args.numerics, max_batch=B, device=dev, **generator_settings(dc))
    ckpt = os.path.join(dc.save_pth, "ldm-last.pt")
    if os.path.exists(ckpt):
        print("Loading Pretrained Models!")
        pipe.load_checkpoint(ckpt)
    else:
        print(f"{ckpt} not found: using named-seed synthetic weights")
        pipe.load_named_seed("shapenet")
    pipe.finalize()
    out_dir = os.path.join(dc.save_pth, "results" if args.mode == "gen" else os.path.join("results", "eval"))
    os.makedirs(out_dir, exist_ok=True)
    if args.mode not in ("gen", "eval"):
        raise SystemExit("data.mode must be gen or eval (training is out of scope)")
    total = world * B if args.mode == "gen" else int(os.environ.get("DDMI_EVAL_SAMPLES", 5000)) // (world * B) * (world * B)
    print("Generating shape!" if args.mode == "gen" else f"Generating {total} shapes for evaluation!")
    for start in range(rank * B, total, world * B):       # disjoint global sample indices per rank
        idx = list(range(start, start + B))
        for i, mesh in zip(idx, pipe.generate(idx, base_seed=args.seed)):
            mesh.export(os.path.join(out_dir, f"generation-{i}.obj"))
    print("Finished generating shapes!")


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--exp", type=str, required=True, choices=["d2c-vae", "ldm"])
    parser.add_argument("--configs", type=str)
    parser.add_argument("--seed", type=int, default=777)
    parser.add_argument("--numerics", type=str, default="bf16", choices=["bf16", "fp32"])
    main(parser.parse_args())
