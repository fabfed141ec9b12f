#!/usr/bin/env python3
"""Benchmark: 512x512 images/s for 50-step DDPM sampling of the SD-1.5-architecture U-Net on MI355X.

    python bench.py                       # 1 GPU, config 2 (batch 1, 64x64 latent, fp32), K=3, W=1
    python bench.py --config 3            # batch 16, bf16
    python bench.py --gpus N              # starts its own N ranks (one fresh process per GPU), or under a launcher:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one complete sampling pass: ``DDPMScheduler(num_train_timesteps=50).sample`` of one batch of
synthetic inputs (random-init deterministic weights, seeded context, device-side noise).  With N > 1
every rank samples its own batch shard (weak scaling, no per-step communication) and the final latents
are all-gathered over RCCL inside the timed region.  Rank 0 prints one JSON line.

The default invocation (config 2, one GPU, no other flags) also runs two side measurements after the headline, each
in a fresh child process, and reports them under ``secondary`` in the same line: config 3 (bf16, batch 16, CLIP
context) and config 2 in the opt-in three-term split-bf16 product mode.  The headline fields are never taken from them.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from diffusion_model_amd.models import CLIP, Diffusion  # noqa: E402
from diffusion_model_amd.parallel import gather_latents, init_distributed, shard_bounds  # noqa: E402
from diffusion_model_amd.schedulers import DDPMScheduler  # noqa: E402
from diffusion_model_amd.schedulers.base_scheduler import sinusoidal_table  # noqa: E402
from diffusion_model_amd.weights import deterministic_fill_  # noqa: E402

CONFIGS = {
    2: dict(batch=1, latent=64, dtype="f32", name="config2: SD-1.5 U-Net, 64x64 latent (512^2), batch 1, 50 DDPM steps, fp32"),
    3: dict(batch=16, latent=64, dtype="bf16", name="config3: SD-1.5 U-Net, 64x64 latent (512^2), batch 16, 50 DDPM steps, bf16"),
    # config 4: ONE job of 128 images; rank r samples images shard_bounds(128, N, r) (64 / 32 / 16 per GPU at
    # N = 2 / 4 / 8), the latents are all-gathered inside the timed region.  Total work is fixed: strong scaling.
    4: dict(batch=128, latent=64, dtype="bf16", sharded=True,
            name="config4: SD-1.5 U-Net, 64x64 latent (512^2), global batch 128 sharded over the GPUs, 50 DDPM steps, bf16"),
    # config 5: the whole text-to-image chain at 768x768 (CLIP encode -> 50 DDPM steps at the 96x96 latent -> VAE decode
    # -> uint8); the value is 768x768 images/s.  fp32 = the reference's precision; --dtype bf16 runs the bf16 kernels.
    5: dict(batch=1, latent=96, dtype="f32", e2e=True,
            name="config5: end-to-end txt2img 768x768: CLIP encode + 50 DDPM steps (96x96 latent) + VAE decode + uint8"),
}
# dense MFMA peaks of MI355X (TFLOP/s): exact-f32 MFMA = vector fp32 rate; bf16 dense (no sparsity)
PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0}
HBM_PEAK_GBS = 8000.0


def build_model(device, dtype):
    with torch.device("meta"):
        model = Diffusion()
    model = model.to_empty(device=device)
    deterministic_fill_(model, seed=0)
    if dtype == "bf16":
        model = model.bfloat16()
    return model.eval()


def kernel_class_profile(model, ctx, latent_hw, batch, device):
    """Eager forward with every launch bracketed by HIP events on the launch stream -> per-class totals."""
    eng, _ = model.hip_pack(device)
    eng.autotune = True
    x = torch.randn(batch, 4, latent_hw, latent_hw, device=device)
    temb = sinusoidal_table(torch.full((batch,), 25), 320).to(device)
    model(x, ctx, temb)  # warm
    model(x, ctx, temb)
    torch.cuda.synchronize(device)
    eng.start_profile()
    model(x, ctx, temb)
    return eng.stop_profile()


def cpu_baseline(latent_hw):
    """The CPU oracle (fp32 restatement of the reference) timed on this host's cores, bounded sample."""
    from oracle import unet_ref
    from diffusion_model_amd.weights import fill_state
    # every core this process may use: the affinity mask, further limited by OMP_NUM_THREADS when the host sets
    # it (a shared GPU box hands each job a CPU share; os.cpu_count() there is the whole machine)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = min(avail, int(os.environ.get("OMP_NUM_THREADS", "0")) or avail)
    torch.set_num_threads(threads)
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    state = fill_state(unet_ref.diffusion_shapes(), seed=0)
    x = torch.randn(1, 4, latent_hw, latent_hw, generator=torch.Generator().manual_seed(3))
    ctx = torch.randn(1, 77, 768, generator=torch.Generator().manual_seed(1))
    temb = sinusoidal_table(torch.tensor([25]), 320)
    unet_ref.diffusion(state, x, ctx, temb)  # warm-up
    n = 3
    t0 = time.perf_counter()
    for _ in range(n):
        unet_ref.diffusion(state, x, ctx, temb)
    fwd = (time.perf_counter() - t0) / n
    return {"value": 1.0 / (50 * fwd), "unit": "images/s", "cores": threads, "kind": "port",
            "cpu_model": cpu_model, "host_logical_cpus": os.cpu_count(),
            "sample": f"{n} timed Diffusion forwards of the CPU oracle at batch 1, {latent_hw}x{latent_hw} latent, fp32 "
                      f"(+1 warm-up): {fwd:.3f} s/forward; images/s extrapolated as 1/(50 x forward)"}


def pmc_traffic(config, batch, latent):
    """HBM bytes per conv3x3 launch from the committed PMC passes of this workload (tools/pmc_traffic.py);
    None when the run is not one of the profiled configurations."""
    name = {2: "pmc_traffic_config2_f32.json", 3: "pmc_traffic_config3_bf16.json"}.get(config)
    if name is None or batch != CONFIGS[config]["batch"] or latent != CONFIGS[config]["latent"]:
        return None
    for rnd in ("r03_", "r02_"):  # the newest committed pass of the workload
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", rnd + name)
        if os.path.exists(path):
            with open(path) as f:
                return round(json.load(f)["per_launch"].get("conv3x3", 0.0)) or None
    return None


def launch_ranks(n: int) -> int:
    """``--gpus n`` outside a launcher: start n fresh child processes of this script, one rank per GPU, and return the
    worst exit code.  This parent has made no GPU call (importing torch makes none); rank 0 inherits stdout, so its
    JSON line is the job's; the other ranks' stdout goes to stderr."""
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    worst = 0
    for pr in procs:
        rc = pr.wait()
        if rc != 0 and worst == 0:
            worst = rc if rc > 0 else 128 - rc
    return worst


def run_stub(args) -> None:
    """Plumbing rehearsal without a GPU (tests only; never a measurement): every rank "samples" its shard of a tiny
    latent batch with torch on the CPU, then the real gather, barriers, max-over-ranks timing and JSON line run."""
    rank, world, _ = init_distributed()
    total = 2 * world + 1  # ragged on purpose
    lo, hi = shard_bounds(total, world, rank)
    full = torch.arange(total * 4 * 2 * 2, dtype=torch.float32).reshape(total, 4, 2, 2)

    def one_step():
        return gather_latents(full[lo:hi] * 2.0 + 1.0, total=total)

    for _ in range(args.warmup):
        one_step()
    if world > 1:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = one_step()
    if world > 1:
        torch.distributed.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tt.item())
    ok = bool(torch.equal(out, full * 2.0 + 1.0))
    if rank == 0:
        print(json.dumps({"metric": "stub (plumbing rehearsal, not a measurement)", "value": args.steps * total / elapsed,
                          "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": elapsed / args.steps * 1e3, "gather_ok": ok,
                          "config": {"workload": "stub", "global_batch": total, "parallelism": f"dp{world}"}}))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if not ok:
        sys.exit(1)
    if os.environ.get("DM_BENCH_STUB_FAIL_RANK") == str(rank):  # lets a test see a rank's failure reach the launcher
        sys.exit(3)


def secondary_run(extra, timeout_s=280):
    """One side measurement in a fresh child process (its own engine and arithmetic mode); returns the fields of its
    JSON line that ``secondary`` reports, or {"error": ...}.  Never raises: the headline line must still be printed."""
    cmd = [sys.executable, os.path.abspath(__file__), "--no-cpu-baseline", "--no-secondary", *extra]
    try:
        t0 = time.perf_counter()
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s)
        wall = time.perf_counter() - t0
        line = next((l for l in reversed(res.stdout.splitlines()) if l.startswith("{")), None)
        if res.returncode != 0 or line is None:
            return {"error": f"exit code {res.returncode}", "stderr_tail": res.stderr[-400:]}
        d = json.loads(line)
        keep = ("value", "unit", "steps", "warmup", "ms_per_step", "dtype", "config", "roofline", "kernel_classes", "setup_s")
        out = {k: d[k] for k in keep if k in d}
        out["command"] = "python bench.py " + " ".join(cmd[2:])
        out["process_wall_s"] = round(wall, 1)
        return out
    except Exception as e:  # timeout, bad JSON, ...
        return {"error": f"{type(e).__name__}: {e}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the config's)")
    ap.add_argument("--dtype", default="", choices=["", "f32", "bf16"], help="override the config's arithmetic type")
    ap.add_argument("--ddpm-steps", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--sampler", default="ddpm", choices=["ddpm", "k_euler_cfg"],
                    help="k_euler_cfg: the scripted loop of the reference's inference.py - 50 K-Euler steps with "
                         "classifier-free guidance (model batch doubled); a side measurement, not the headline metric")
    ap.add_argument("--fp32-split", type=int, nargs="?", const=2, default=0, choices=[0, 2, 3],
                    help="side measurement, NOT the headline metric: the fp32 configurations with the engine's opt-in "
                         "split-bf16 product modes (DM_FP32_SPLIT): 2 = two bf16 terms per operand, three products, about "
                         "16 bits of each operand (GEMM, conv and attention); 3 = three terms, six products, 24 bits "
                         "(GEMM and conv; attention exact fp32); the line's dtype field says which")
    ap.add_argument("--no-secondary", action="store_true",
                    help="default invocation only: skip the config-3 and split-3 side measurements")
    ap.add_argument("--stub", action="store_true", help=argparse.SUPPRESS)  # CPU rehearsal of the multi-rank plumbing
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under torchrun: this process becomes the launcher of its own ranks and touches no GPU itself
        sys.exit(launch_ranks(args.gpus))
    if args.stub:
        return run_stub(args)
    # the plain default invocation (what the round-end driver runs) also reports the side measurements
    default_run = (args.gpus == 1 and args.config == 2 and not args.batch and not args.dtype and args.ddpm_steps == 50
                   and args.sampler == "ddpm" and not args.fp32_split and not args.no_graph and not args.no_secondary
                   and int(os.environ.get("WORLD_SIZE", "1")) == 1)
    if args.fp32_split:
        os.environ["DM_FP32_SPLIT"] = str(args.fp32_split)  # read when the engine is created

    cfg = dict(CONFIGS[args.config])
    if args.batch:
        cfg["batch"] = args.batch
    if args.dtype:
        cfg["dtype"] = args.dtype
        cfg["name"] += f" [--dtype {args.dtype}]"
    rank, world, local = init_distributed()
    global_batch = cfg["batch"] if cfg.get("sharded") else cfg["batch"] * world
    if cfg.get("sharded"):
        lo, hi = shard_bounds(cfg["batch"], world, rank)
        cfg["batch"] = hi - lo
    if world != args.gpus:
        if rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    device = torch.device("cuda", local % max(torch.cuda.device_count(), 1))  # ranks share GPUs only in rehearsals
    torch.cuda.set_device(device)

    B, HW, T = cfg["batch"], cfg["latent"], args.ddpm_steps
    t_setup = time.perf_counter()
    model = build_model(device, cfg["dtype"])
    decoder = None
    if cfg.get("e2e"):
        from diffusion_model_amd.models import Decoder
        from diffusion_model_amd.pipeline import txt2img_ddpm
        decoder = deterministic_fill_(Decoder().to(device), seed=0)
        decoder = (decoder.bfloat16() if cfg["dtype"] == "bf16" else decoder).eval()
    if args.config in (3, 4, 5):
        # config 3 conditions on CLIP output: random-init text encoder of the same dtype, random token ids
        # (seed 2 + rank); the encode runs once per sampling pass, inside the timed region
        with torch.device("meta"):
            clip = CLIP()
        clip = deterministic_fill_(clip.to_empty(device=device), seed=0)
        clip = (clip.bfloat16() if cfg["dtype"] == "bf16" else clip).eval()
        if cfg.get("sharded"):  # one global prompt table, every rank takes the rows of its images
            tokens = torch.randint(0, 49408, (global_batch, 77), generator=torch.Generator().manual_seed(2))[lo:hi].to(device)
        else:
            tokens = torch.randint(0, 49408, (B, 77), generator=torch.Generator().manual_seed(2 + rank)).to(device)
        ctx = None
    else:
        clip, tokens = None, None
        ctx = torch.randn(B, 77, 768, generator=torch.Generator().manual_seed(1 + rank)).to(device)
    sched = DDPMScheduler(num_train_timesteps=T, clip_sample=True)
    sched.use_hip_graph = not args.no_graph
    gen = torch.Generator(device=device).manual_seed(42 + rank)
    shape = (B, 4, HW, HW)

    if args.sampler == "k_euler_cfg":
        from diffusion_model_amd.pipeline import denoise, initial_latents
        from diffusion_model_amd.sampling import KEulerSampler
        cfg["name"] += " [K-Euler, classifier-free guidance 7.5, model batch doubled]"
        ksampler = KEulerSampler(n_inference_steps=T)
        ctx_pair = (ctx if ctx is not None else clip(tokens).float())
        ctx_pair = torch.cat([ctx_pair, torch.zeros_like(ctx_pair)])  # [conditional | unconditional]

    def one_step():
        if args.sampler == "k_euler_cfg":
            ksampler.reset()
            lat = initial_latents(ksampler, shape, gen, device)
            return denoise(model, ksampler, ctx_pair, lat, cfg_scale=7.5, do_cfg=True, use_hip_graph=not args.no_graph)
        if decoder is not None:  # tokens -> uint8 pixels, everything on the device
            pixels, _ = txt2img_ddpm(tokens, clip, model, decoder, sched, 8 * HW, 8 * HW, gen)
            return pixels
        context = clip(tokens).float() if clip is not None else ctx
        lat = sched.sample(model, shape, context, gen, device)
        return gather_latents(lat, total=global_batch) if world > 1 else lat

    # set-up, outside warm-up and timing: weight packing and the per-shape GEMM tile/split autotune run on the
    # first forward of a process (like a compile step); a 2-step sampling call triggers both
    setup = DDPMScheduler(num_train_timesteps=2, clip_sample=True)
    setup.use_hip_graph = False
    setup.sample(model, shape, clip(tokens).float() if clip is not None else ctx, gen, device)
    del setup
    if decoder is not None:
        decoder(torch.zeros(shape, device=device))  # packs the decoder's weights
    torch.cuda.synchronize(device)
    setup_s = time.perf_counter() - t_setup
    for _ in range(args.warmup):
        one_step()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = one_step()
    torch.cuda.synchronize(device)
    if world > 1:
        torch.distributed.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert out.dtype == torch.uint8 or torch.isfinite(out).all()

    if rank == 0:
        prof = kernel_class_profile(model, clip(tokens).float() if clip is not None else ctx, HW, B, device)
        conv = prof["conv3x3"]
        tflops = conv["work"] / (conv["ms"] * 1e-3) / 1e12
        peak = PEAK_TFLOPS[cfg["dtype"]]
        split = args.fp32_split if cfg["dtype"] == "f32" else 0
        if split:
            peak = PEAK_TFLOPS["bf16"] / (3 if split == 2 else 6)  # bf16 products per fp32 product (conv / GEMM classes)
        # every kernel class of one U-Net forward against its roofline: MFMA classes (algorithmic 2*M*N*K flops) vs the
        # dense MFMA peak of the dtype, the others (algorithmic bytes: one read + one write) vs 8 TB/s of HBM
        classes = {}
        for k, v in prof.items():
            rate = v["work"] / (v["ms"] * 1e-3)
            mfma = k in ("conv3x3", "gemm", "attention")
            classes[k] = {"ms_per_forward": round(v["ms"], 4), "launches": v["launches"],
                          ("tflops" if mfma else "gbs"): round(rate / (1e12 if mfma else 1e9), 2),
                          "bound": "mfma" if mfma else "hbm",
                          "frac": round(rate / (peak * 1e12 if mfma else HBM_PEAK_GBS * 1e9), 4)}
        result = {
            "metric": ("768x768 images/sec, end-to-end txt2img (CLIP + 50-step DDPM + VAE decode)" if cfg.get("e2e")
                       else "512x512 images/sec, 50-step DDPM, SD-1.5 U-Net"),
            "value": args.steps * global_batch / elapsed,
            "unit": "images/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if cfg.get("sharded") else "weak",
            "vs_baseline": None,
            "dtype": (cfg["dtype"] if not split else
                      "f32 operands as 2 x bf16 (3 products, fp32 accumulation): reduced precision, side measurement" if split == 2
                      else "f32 operands as 3 x bf16 (6 products, fp32 accumulation; attention exact f32): side measurement"),
            "data": "synthetic (deterministic random-init weights, "
                    + ("CLIP context from random token ids, " if clip is not None else "seeded random context, ")
                    + "device-generated noise)",
            "config": {"workload": cfg["name"], "batch_per_gpu": B, "global_batch": global_batch,
                       "latent": f"{HW}x{HW}", "ddpm_steps": T, "hip_graph": sched.use_hip_graph,
                       "parallelism": f"dp{world}"},
            "roofline": {"bound": "mfma", "kernel": "conv3x3_igemm_kernel + conv3x3_pair_kernel"
                                                    + (" + conv3x3_chunk_kernel + conv3x3_halo_kernel + conv3x3_wide_kernel" if cfg["dtype"] == "bf16"
                                                       else " + conv3x3_gn_kernel" if not split else " (conv3x3_split_kernel)"),
                         "achieved": round(tflops, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(tflops / peak, 4), "traffic": pmc_traffic(args.config, B, HW),
                         "traffic_unit": "bytes/launch (rocprofv3 FETCH_SIZE*2 + WRITE_SIZE, separate passes)",
                         "launches_per_forward": conv["launches"],
                         "avg_launch_us": round(conv["ms"] * 1e3 / conv["launches"], 2)},
            "kernel_classes": classes,
            "setup_s": round(setup_s, 1),
            "setup": "model build + deterministic fill, weight packing, first-use tile selection for shapes missing from "
                     "the committed tile table; outside warm-up and the timed region",
        }
        if not args.no_cpu_baseline and world == 1:
            result["cpu_baseline"] = cpu_baseline(HW)
        if default_run:
            # side measurements under the same clock as the headline, each in a fresh process; they never replace a
            # headline field.  config3 = the bf16 configuration the MFMA targets are quoted on; config2_split3 = this
            # workload with every fp32 operand carried as three bf16 terms (24 bits, six products) - not the
            # reference's arithmetic, reported for comparison only.
            result["secondary"] = {
                "config3": secondary_run(["--config", "3", "--steps", "3", "--warmup", "1"]),
                "config2_split3": secondary_run(["--config", "2", "--fp32-split", "3", "--steps", str(args.steps),
                                                 "--warmup", str(args.warmup)]),
            }
        print(json.dumps(result))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
