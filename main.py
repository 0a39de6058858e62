#!/usr/bin/env python3
"""Entry point with the reference's surface: `python main.py --exp ldm --configs configs/ldm/afhq.yaml [--seed 777]`
(reference main.py:54-65), also under `accelerate launch` / `torch.distributed.run` (one process per GPU).

The generation path (`data.mode: gen | eval` of the `ldm` experiment; image, video and occupancy domains) and the image-domain
LDM training loop on latents (`data.mode: train`, ddmi_amd/train.py; image, occupancy and video domains) are implemented;
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
    if args.mode == "train":
        return train_main(args, dc, rank, world, dev)
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
    load_weights(pipe, args, dc, "sky" if args.domain == "video" else "afhq")
    pipe.finalize()
    out_dir = os.path.join(dc.save_pth, "results")
    os.makedirs(out_dir, exist_ok=True)
    gen = lambda idx, first: pipe.generate(idx, base_seed=args.seed, rng=args.rng, reseed=first)
    if args.mode == "gen":
        print("Random image generation!")
        img = D.generate_sharded(lambda idx: gen(idx, True), world * B, B, rank, world)
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
        # evals/eval.py:215-234: `total_fake_number // batch_size` batches of samples, each written out as uint8.  Every
        # batch is converted to uint8 ON THE DEVICE (ddmi_image_to_u8, bit-identical to the numpy expression) and only
        # those bytes are gathered and written by rank 0 as one shard per global batch - fp32 samples never accumulate
        # (10k images would be 7.9 GB per rank, 10k video clips 126 GB).
        from ddmi_amd.mcubes import images_to_uint8
        total = int(os.environ.get("DDMI_EVAL_SAMPLES", 10000)) // (world * B) * (world * B)
        fid_dir = os.path.join(out_dir, "fid50k")   # tools/ldm/image.py:221
        os.makedirs(fid_dir, exist_ok=True)
        mine = D.shard_indices(total, rank, world)
        for k in range(0, len(mine), B):
            idx = mine[k:k + B]
            img = gen(idx, k == 0)
            if args.domain == "video":   # [B,3,T,H,W] -> frames as images [B*T,3,H,W]
                Bv, Cv, Tv, Hv, Wv = img.shape
                u8 = images_to_uint8(img.permute(0, 2, 1, 3, 4).reshape(Bv * Tv, Cv, Hv, Wv).contiguous()).reshape(Bv, Tv, Hv, Wv, Cv)
            else:
                u8 = images_to_uint8(img)
            full = D.reorder_rank_major(D.all_gather_samples(u8, world), world, len(idx))
            if rank == 0:
                np.save(os.path.join(fid_dir, f"gen-{k // B:05d}.npy"), full.cpu().numpy())
        if rank == 0:
            print(f"wrote {total} samples to {fid_dir}")
    else:
        raise SystemExit("data.mode must be gen, eval or train")


def load_weights(pipe, args, dc, tag):
    """The reference's three start modes (tools/ldm/image.py:70-88): resume / pretrained -> ldm-last.pt, otherwise the
    stage-1 model-last.pt; without any checkpoint on disk, named-seed synthetic weights.  A checkpoint fixes the length of
    the noise schedule it was trained with: when the YAML's `timesteps` differs (e.g. the 250-step benchmark config), the
    checkpoint's T is used and `sampling_timesteps` is kept, so DDPM.sample takes the DDIM path as the reference would."""
    from ddmi_amd import checkpoint as CK
    from ddmi_amd.ddpm import DDPM
    path = CK.find_checkpoint(dc.save_pth, resume=bool(getattr(args, "resume", False)), pretrained=bool(args.pretrained))
    if path is None:
        print(f"no checkpoint under {dc.save_pth}: using named-seed synthetic weights")
        pipe.load_named_seed(tag)
        return
    print("Loading Pretrained Models!" if path.endswith("ldm-last.pt") else "Load VAE checkpoints!")
    ck = CK.read_checkpoint(path)    # once: the dict is handed to load_checkpoint below
    T_ck = CK.checkpoint_sampler_config(ck)
    if T_ck and T_ck["timesteps"] != pipe.ddpm.num_timesteps:
        cfg = dict(args.ddpmconfig)
        print(f"checkpoint schedule has T={T_ck['timesteps']} (config: {cfg.get('timesteps')}): sampling "
              f"{cfg.get('sampling_timesteps')} steps over the checkpoint's schedule")
        cfg["timesteps"] = T_ck["timesteps"]
        if cfg.get("sampling_timesteps") is None or cfg["sampling_timesteps"] > cfg["timesteps"]:
            cfg["sampling_timesteps"] = cfg["timesteps"]
        if args.domain == "video":
            cfg["domain"] = "video"
        else:
            cfg.pop("domain", None)
        pipe.ddpm = DDPM(model=pipe.unet, **cfg)
    if path.endswith("model-last.pt"):
        print("stage-1 checkpoint: the diffusion model keeps its initialisation (named-seed weights)")
        pipe.load_named_seed(tag)
    pipe.load_checkpoint(ck)


def train_main(args, dc, rank, world, dev):
    """`data.mode: train` (image, occupancy and video domains): the LDM trainer's loop (tools/ldm/image.py:136-178, tools/ldm/occupancy.py:139-180, tools/ldm/video.py:125-167) on LATENTS - the VAE encoder that
    produces them from images is outside this path.  Latents come from `data.latents` (an .npy of [N,C,H,W] fp32, e.g.
    `vaemodel.encode(y).sample()` dumped by the reference) or, without one, are synthetic N(0,1).  One process per GPU; every
    rank draws its own batches and the gradient slab is all-reduced over RCCL each optimizer step.  Writes `ldm-last.pt` in the
    layout LDMTrainer.save uses, so `mode: gen` picks it up."""
    import numpy as np
    import torch
    from ddmi_amd import checkpoint as CK
    from ddmi_amd.train import LDMTrainStep
    # image (tools/ldm/image.py:136-178) and occupancy (tools/ldm/occupancy.py:139-180: the same step on z = cat(z_xy, z_xz, z_yz), a
    # [B,192,16,16] latent through the same 2-D UNetModel); video (tools/ldm/video.py:150-167): z = cat(z_xy, z_xt, z_yt) [B, C, L]
    # through UNetModel_Triplane (ddmi_amd/train.py TrainUNetModel_Triplane), mixing_logit [1, C, 1]
    if args.domain not in ("image", "occupancy", "video"):
        raise SystemExit("mode: train is implemented for the image, occupancy and video domains")
    video = args.domain == "video"
    lc = args.loss_config or {}
    B = int(dc.get("batch_size", dc.test_batch_size))
    steps = int(os.environ.get("DDMI_TRAIN_STEPS", dc.get("train_steps", 100)))
    step = LDMTrainStep(args.unetconfig, dict(args.ddpmconfig, domain="video") if video else args.ddpmconfig, numerics=args.numerics,
                        lr=float(getattr(args, "lr", None) or 1e-4),
                        gradient_accumulate_every=int(lc.get("gradient_accumulate_every", 1)), ema_decay=float(lc.get("ema_decay", 0.9999)),
                        ema_update_every=int(lc.get("ema_update_every", 10)), device=dev)
    vae_state, mlp_state, ck = {}, {}, None
    path = CK.find_checkpoint(dc.save_pth, resume=bool(getattr(args, "resume", False)), pretrained=bool(args.pretrained))
    if path is not None:   # the reference's start modes: resume / pretrained continue an LDM, otherwise the stage-1 VAE + MLP
        ck = CK.read_checkpoint(path)
        T_ck = CK.checkpoint_sampler_config(ck)
        if T_ck and T_ck["timesteps"] != int(args.ddpmconfig.get("timesteps", 1000)):
            raise SystemExit(f"checkpoint was trained with T={T_ck['timesteps']}, the config says {args.ddpmconfig.get('timesteps')}: "
                             "continuing it on another schedule is not the reference's resume")
        st = CK.sampling_states(ck, use_ema=False)
        vae_state, mlp_state = st["vae"], st["mlp"]
        if st["diffusion"] is not None:
            step.load_state_dict({k: v for k, v in st["diffusion"].items() if k.startswith("model.") or k == "mixing_logit"}, strict=True)
            print("Loading Models from previous training!" if getattr(args, "resume", False) else "Loading Pretrained Models!")
    if path is None or not path.endswith("ldm-last.pt"):
        from ddmi_amd import initweights as iw   # a fresh diffusion model: named-seed init (deterministic across ranks)
        tag = {"image": "afhq.unet", "occupancy": "shapenet.unet", "video": "sky.unet3"}[args.domain]
        st = iw.fill_state(tag, [(k, tuple(p.shape)) for k, p in step.model.state_dict().items()])
        step.load_state_dict({"model." + k: v for k, v in st.items()}, strict=False)
    step.finalize()
    if ck is not None and path.endswith("ldm-last.pt"):
        # tools/ldm/image.py:76-82 (pretrained: + EMA) and :121-134 (resume: + optimizer moments, step counters): without this the
        # EMA would restart from the online weights and AdamW's bias correction from step 0
        step.load_training_state(ck, resume=bool(getattr(args, "resume", False)))
        print(f"restored EMA (step {step.ema_step})" + (f", AdamW state, optimizer step {step.opt_step}" if getattr(args, "resume", False) else ""))
    lat = None
    if dc.get("latents") and os.path.exists(dc.latents):
        lat = torch.from_numpy(np.load(dc.latents).astype(np.float32))
    g = torch.Generator(device=dev).manual_seed(args.seed * 1000 + rank)
    C_, S_ = int(args.ddpmconfig.get("channels", 64)), int(args.ddpmconfig.get("image_size", 64))
    for it in range(steps):
        if lat is not None:
            idx = torch.randint(0, lat.shape[0], (B,), generator=torch.Generator().manual_seed(args.seed * 7919 + it * world + rank))
            z = lat[idx].to(dev)
        else:
            z = torch.randn((B, C_, step.model.seq_len) if video else (B, C_, S_, S_), device=dev, generator=g)
        loss = step.train_step(z, world_size=world)
        if rank == 0 and (it % 10 == 0 or it == steps - 1):
            print(f"iter {it}: Dae loss : {float(loss):.4f}", flush=True)
    if rank == 0:
        os.makedirs(dc.save_pth, exist_ok=True)
        plain = lambda d: {k: v for k, v in dict(d or {}).items() if isinstance(v, (int, float, str, bool, list, type(None)))}
        step.save(os.path.join(dc.save_pth, "ldm-last.pt"), vae_state, mlp_state,
                  args={"unetconfig": plain(args.unetconfig), "ddpmconfig": plain(args.ddpmconfig), "loss_config": plain(lc),
                        "lr": step.lr, "embed_dim": int(args.embed_dim)})
        print(f"saved {os.path.join(dc.save_pth, 'ldm-last.pt')} after {step.opt_step} optimizer steps")


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
    load_weights(pipe, args, dc, "shapenet")
    pipe.finalize()
    out_dir = os.path.join(dc.save_pth, "results" if args.mode == "gen" else os.path.join("results", "eval"))
    os.makedirs(out_dir, exist_ok=True)
    if args.mode not in ("gen", "eval"):
        raise SystemExit("data.mode must be gen or eval (training is out of scope)")
    total = world * B if args.mode == "gen" else int(os.environ.get("DDMI_EVAL_SAMPLES", 5000)) // (world * B) * (world * B)
    print("Generating shape!" if args.mode == "gen" else f"Generating {total} shapes for evaluation!")
    for start in range(rank * B, total, world * B):       # disjoint global sample indices per rank
        idx = list(range(start, start + B))
        for i, mesh in zip(idx, pipe.generate(idx, base_seed=args.seed, rng=args.rng, reseed=(start == rank * B))):
            mesh.export(os.path.join(out_dir, f"generation-{i}.obj"))
    print("Finished generating shapes!")


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--exp", type=str, required=True, choices=["d2c-vae", "ldm"])
    parser.add_argument("--configs", type=str)
    parser.add_argument("--seed", type=int, default=777)
    parser.add_argument("--numerics", type=str, default="bf16", choices=["bf16", "fp32"])
    parser.add_argument("--rng", type=str, default="philox", choices=["philox", "reference"],
                        help="philox: per-sample counter-based streams (sharding-invariant); reference: torch.manual_seed(--seed) "
                             "and CPU randn in the reference's draw order (same samples as the reference's CPU path)")
    main(parser.parse_args())
