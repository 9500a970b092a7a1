#!/usr/bin/env python3
"""SAE train-step throughput on MI355X: activations/sec through the TopK-SAE train step
(d = 384 -> 3072, k = 32), BASELINE.json's metric on its config 2 (1 GPU) / 3 (N GPUs).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one full ``SAETrainer.train_step`` on a batch of B activation rows already resident in the
on-device ring buffer: stage + encode GEMM + TopK + sparse decode/MSE + weight-gradient GEMMs +
(RCCL all-reduce for N > 1) + clip + AdamW + decoder renorm + dead-feature scan.  Weak scaling:
B per GPU fixed, global batch = N * B.  Rank 0 prints ONE JSON line.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
for _p in (str(ROOT), str(ROOT / "whisper-sae_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

D_MODEL, HIDDEN, TOPK = 384, 3072, 32
BF16_DENSE_PEAK_TFLOPS = 2500.0  # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md, chip-level parameters)
HBM_PEAK_GBS = 8000.0
PMC_TRAFFIC_FILE = "r03_pmc_traffic.json"  # written by profiles/tools/profile_step.sh on this build


def _cpu_leg(batch: int, budget_s: float, max_steps: int, cores: int) -> tuple:
    import torch

    from oracle import synth
    from oracle.torch_step import TorchCPUStep

    w = synth.sae_weights(D_MODEL, HIDDEN, seed=42, bf16=False)
    step = TorchCPUStep(w, TOPK, lr=1e-4, weight_decay=0.0, max_norm=1.0)
    x = torch.from_numpy(synth.activations(batch, D_MODEL, seed=42, stream=0, bf16=False))
    step.step(x)  # warm-up (allocations, thread pool)
    t0 = time.perf_counter()
    n = 0
    while True:
        step.step(x)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= max_steps:
            break
    return n, el


def cpu_baseline(batch: int, budget_s: float = 16.0) -> dict:
    """Reference-semantics CPU train step (oracle/torch_step.py, kind "port") on the host cores: the headline batch
    (``value``) and, as SURVEY.md section 8 row D asks, the reference YAMLs' own B = 128 and B = 4096 beside it."""
    import torch

    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    # a 1-GPU box gives this process a 16-core share of the host; oversubscribing the 256 visible
    # hardware threads made torch ~10x slower.  WSAE_CPU_THREADS overrides.
    cores = int(os.environ.get("WSAE_CPU_THREADS", min(cores, 16)))
    torch.set_num_threads(cores)
    n, el = _cpu_leg(batch, budget_s, 50, cores)
    others = []
    for b, budget, cap in ((128, 3.0, 400), (4096, 5.0, 60)):
        if b == batch:
            continue
        nb, elb = _cpu_leg(b, budget, cap, cores)
        others.append({"batch": b, "value": b * nb / elb, "steps": nb, "seconds": elb})
    return {"value": batch * n / el, "unit": "activations/s", "cores": cores, "kind": "port",
            "sample": f"{n} train steps of B={batch} (384->3072, k=32, fp32, torch {torch.__version__} CPU, "
                      f"{el:.1f} s); oracle/torch_step.py restates training.py:161-217",
            "other_batches": others}


# Algorithmic work per launch of the step's kernels at batch B: flops on MFMA and the bytes the
# kernel cannot avoid moving given what the step hands it (inputs read once, outputs written once).
def kernel_work(name: str, B: int, D: int, H: int, K: int) -> dict:
    P = 2 * D * H + H + 2 * D
    if name.startswith("encode_gemm"):
        return {"flops": 2.0 * B * D * H, "bytes": 2.0 * B * D + 2.0 * H * D + 4.0 * B * H + 4.0 * B * H / 16,
                "bound": "hbm-store", "what": "x rows + bf16 W_e in, fp32 pre [B,H] + strip maxima out"}
    if name.startswith("wgrad_reduce") or name.startswith("grad_finish"):
        return {"flops": 0.0, "bytes": 4.0 * 8 * 2 * H * D + 4.0 * P, "bound": "hbm", "what": "8 slabs in, gradient pack out"}
    if name.startswith("wgrad"):
        return {"flops": 2 * (2.0 * H * D * B), "bytes": 2 * 2.0 * B * D + 12.0 * B * K + 4.0 * 8 * 2 * H * D,
                "bound": "mfma", "what": "g and x rows + bucketed code in, 8 split-K slabs out"}
    if name.startswith("decode"):
        return {"flops": 2 * (2.0 * B * K * D), "bytes": 2.0 * B * D + 8.0 * B * K + 2.0 * B * D + 4.0 * B * K + 2.0 * H * D,
                "bound": "hbm", "what": "x rows + code in, bf16 g + dpre out, bf16 W_d (L2-resident gathers: 2 B K D bytes)"}
    if name.startswith("topk"):
        return {"flops": 0.0, "bytes": 4.0 * B * H / 16 + 4.0 * B * H * 0.22 + 8.0 * B * K, "bound": "hbm",
                "what": "strip maxima + the ~22 % of pre strips at or above the row threshold in, code out"}
    if name.startswith("bucket"):
        return {"flops": 0.0, "bytes": 2 * 12.0 * B * K, "bound": "hbm", "what": "code in, bucketed code out"}
    if name.startswith("adamw") or name.startswith("update_rows"):
        return {"flops": 0.0, "bytes": 28.0 * P + 2.0 * 2 * H * D, "bound": "hbm", "what": "p, g, m, v in; p, m, v + bf16 shadows out"}
    return {"flops": 0.0, "bytes": 0.0, "bound": "hbm", "what": ""}


def roofline_object(name: str, n: int, ms: float, B: int, traffic_rec, launches_per_step: int = 1) -> dict:
    w = dict(kernel_work(name, B, D_MODEL, HIDDEN, TOPK))
    if launches_per_step > 1:  # data parallel with ddp_overlap_halves: each kernel's work splits evenly over its two launches
        w["flops"] /= launches_per_step
        w["bytes"] /= launches_per_step
    t = ms / n * 1e-3
    tf = w["flops"] / t / 1e12
    gbs = w["bytes"] / t / 1e9
    mfma = w["bound"] == "mfma"
    out = {"bound": w["bound"], "kernel": name, "achieved": tf if mfma else gbs,
           "peak": BF16_DENSE_PEAK_TFLOPS if mfma else HBM_PEAK_GBS, "unit": "TFLOP/s" if mfma else "GB/s",
           "frac": (tf / BF16_DENSE_PEAK_TFLOPS) if mfma else (gbs / HBM_PEAK_GBS),
           "avg_launch_ms": ms / n, "launches": n, "flops_per_launch": w["flops"], "algorithmic_bytes_per_launch": w["bytes"],
           "algorithmic_bytes_are": w["what"], "mfma_frac": tf / BF16_DENSE_PEAK_TFLOPS, "hbm_frac": gbs / HBM_PEAK_GBS,
           "traffic": None, "traffic_source": None}
    if traffic_rec:
        out["traffic"], out["traffic_source"] = traffic_rec
    return out


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--batch", type=int, default=16384, help="activation rows per GPU per step")
    ap.add_argument("--ring-rows", type=int, default=1 << 22)
    ap.add_argument("--precision", choices=("bf16", "fp32"), default="bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--windows", type=int, default=9,
                    help="timed windows of --steps steps each; the median is reported (every window is listed: the first ~100 steps after an idle GPU run at a ramping clock)")
    ap.add_argument("--grad-exchange", choices=("auto", "fp32", "bf16"), default="auto",
                    help="N > 1: dtype of the gradient all-reduce; auto (the bench default) = bf16 in the bf16 mode, fp32 in the "
                         "fp32 mode - named in the line's config.workload / config.grad_exchange; the library default "
                         "(TrainingConfig.grad_exchange_dtype) is fp32")
    ap.add_argument("--profile-all", action="store_true", help="time every kernel (adds event overhead)")
    ap.add_argument("--dims", type=int, nargs=3, metavar=("D", "H", "K"), default=None,
                    help="informational: other SAE dimensions (e.g. 768 12288 64 = BASELINE.json configs[3]); no roofline object")
    ap.add_argument("--fp8", action="store_true", help="with --relu: e4m3 operands in the two forward GEMMs (configs[4])")
    ap.add_argument("--relu", action="store_true", help="informational: ReLU+L1 SAE step (row A12) instead of the TopK "
                                                        "headline; no roofline object")
    args = ap.parse_args()

    global D_MODEL, HIDDEN, TOPK
    if args.dims:
        D_MODEL, HIDDEN, TOPK = args.dims

    import torch
    import torch.distributed as dist

    from whisper_sae import _native as N
    from whisper_sae.config import TrainingConfig
    from whisper_sae.data import ActivationRing, RingLoader
    from whisper_sae.sae.model import ReLUSAE, TopKSAE
    from whisper_sae.sae.training import SAETrainer

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N > 1")
    # WSAE_BENCH_REHEARSE=1: walk the N > 1 code on a ONE-GPU box - every rank on cuda:0, gloo instead of RCCL.  The line
    # it prints carries "rehearsal": true and is not a measurement.
    rehearse = world > 1 and os.environ.get("WSAE_BENCH_REHEARSE") == "1"
    if rehearse:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    B = args.batch
    torch.manual_seed(42)  # same initial weights on every rank (scripts/train.py:84-90 seeds before create_sae)
    model = (ReLUSAE(D_MODEL, HIDDEN, sparsity_weight=0.01, precision="fp8" if args.fp8 else None) if args.relu
             else TopKSAE(D_MODEL, HIDDEN, k=TOPK))
    cfg = TrainingConfig(batch_size=B, learning_rate=1e-4, weight_decay=0.0, warmup_steps=1000, gradient_clip=1.0,
                         use_amp=(args.precision == "bf16"), num_workers=0, seed=42,
                         grad_exchange_dtype=args.grad_exchange)
    trainer = SAETrainer(model, cfg, device=device, run_dir=ROOT / "gpurun_out" / f"bench_rank{rank}")
    ring_dtype = torch.bfloat16 if args.precision == "bf16" else torch.float32
    ring = ActivationRing(args.ring_rows, D_MODEL, device=device, dtype=ring_dtype)
    ring.fill_synthetic(args.ring_rows, seed=42 + rank)  # every rank owns its own shard of rows
    loader = RingLoader(ring, B, shuffle=True, seed=42)
    total = args.warmup + (args.windows + 1) * args.steps + 8 + 224
    trainer.setup_scheduler(max(total, 20000))

    def batches():
        while True:
            for b in loader:
                if len(b) == B:
                    yield b

    it = batches()
    for _ in range(args.warmup):
        trainer.train_step(next(it))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    eng = model._engine
    prec = N.PREC_BF16 if args.precision == "bf16" else N.PREC_FP32
    handle = eng.ctx(prec, B)
    # Which kernel is the dominant one is MEASURED, not assumed: an untimed probe window with HIP events around every
    # launch (recorded by the library on the launch stream) ranks the step's kernels by average duration; the timed
    # windows then carry events around that kernel only (two event records per step).
    probe = {}
    n_settle = n_probe = 0
    if not args.profile_all:
        # (the ranking is only meaningful at the clock the timed windows run at: an idle GPU needs ~100 steps to get there, so
        # with a short --warmup the probe is preceded by enough plain steps to make 200 untimed steps in all; the line reports
        # every untimed step)
        n_probe = 24
        n_settle = max(0, 200 - args.warmup - n_probe)
        for _ in range(n_settle):
            trainer.train_step(next(it))
        N.check(N.lib().wsae_profile_enable(handle, -1, n_probe), "wsae_profile_enable")
        for _ in range(n_probe):
            trainer.train_step(next(it))
        probe = N.profile_read(handle)
        N.check(N.lib().wsae_profile_disable(handle), "wsae_profile_disable")
    kid = -1
    dominant = None
    if probe:
        dominant = max(probe, key=lambda k: probe[k][1] / max(probe[k][0], 1))
        kid = next(k for k in range(N.KERNEL_COUNT) if N.lib().wsae_kernel_name(k).decode() == dominant)
    N.check(N.lib().wsae_profile_enable(handle, kid, args.steps * args.windows), "wsae_profile_enable")

    # windows of EXACTLY --steps steps, each bracketed by barrier + synchronize on both
    # sides and reduced with MAX over ranks; the median window is the reported one, all of them are listed.
    last = None
    windows = []
    for _ in range(max(args.windows, 1)):
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            last = trainer.train_step(next(it))
        barrier()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        windows.append(el)
    elapsed = sorted(windows)[len(windows) // 2]

    prof = N.profile_read(handle)
    N.check(N.lib().wsae_profile_disable(handle), "wsae_profile_disable")

    if rank == 0:
        value = world * B * args.steps / elapsed
        # flop per activation: the reference's 2 fwd + 4 bwd dense GEMMs (TopK), 2 + 3 without dL/dx (ReLU, row A12)
        f_dense = (10 if args.relu else 12) * D_MODEL * HIDDEN
        # roofline of the dominant kernel (by measured time), HIP-event duration inside the timed windows; the
        # weight-gradient contraction - the MFMA-bound kernel of the step - keeps an object of its own from the probe window
        roof, roof_wgrad = None, None
        headline = not args.dims and not args.relu
        pmc_k = {}
        pmc = ROOT / "profiles" / PMC_TRAFFIC_FILE
        if headline and B == 16384 and args.precision == "bf16" and pmc.exists():
            pmc_k = json.loads(pmc.read_text())["kernels"]

        pmc_names = {"encode_gemm": "encode_gemm256d_kernel", "topk": "topk_strips_kernel", "decode": "decode_mfma_kernel",
                     "bucket": "bucket_kernel", "wgrad": "wgrad2_kernel", "wgrad_reduce": "grad_finish_kernel",
                     "adamw": "update_rows_kernel"}

        def traffic_of(name):
            # HBM bytes per launch: the FETCH_SIZE / WRITE_SIZE PMC passes of this build committed under profiles/
            # (profiles/tools/profile_step.sh; PMC cannot be collected inside a timed run); valid at that configuration only
            rec = pmc_k.get(pmc_names.get(name, ""))
            if rec:
                return rec["hbm_bytes"], (f"profiles/{PMC_TRAFFIC_FILE} (2 x FETCH_SIZE + WRITE_SIZE, median per launch of "
                                          f"{pmc_names[name]})")
            return None

        if headline:
            names = dict(prof)
            halves = 2 if (world > 1 and cfg.ddp_overlap_halves and N.lib().wsae_wgrad_parts_supported(handle)) else 1

            def lps(name):
                return halves if name in ("wgrad", "wgrad_reduce") else 1

            if dominant and dominant in names and names[dominant][0]:
                roof = roofline_object(dominant, names[dominant][0], names[dominant][1], B, traffic_of(dominant), lps(dominant))
                roof["selected_by"] = "longest average launch in the probe window (HIP events around every kernel)"
                roof["kernel"] = pmc_names.get(dominant, dominant) + f"<{args.precision}>"
            src = prof if "wgrad" in prof else probe
            if "wgrad" in src and src["wgrad"][0]:
                roof_wgrad = roofline_object("wgrad", src["wgrad"][0], src["wgrad"][1], B, traffic_of("wgrad"), lps("wgrad"))
                roof_wgrad["kernel"] = "wgrad2_kernel<bf16>" if args.precision == "bf16" else "wgrad2_kernel<f32>"
                roof_wgrad["measured_in"] = "timed windows" if src is prof else "probe window (events around every kernel)"
            if roof is None:
                roof = roof_wgrad
            if roof is not None:
                roof.update({"step_dense_equiv_tflops": value * f_dense / 1e12,
                             "step_dense_equiv_frac": value * f_dense / 1e12 / BF16_DENSE_PEAK_TFLOPS,
                             # what the step really issues on MFMA: encoder GEMM 2*D*H + the two weight-gradient
                             # contractions 4*D*H per activation (the decode and dpre products are sparse, k rows each)
                             "f_exec_per_activation": 6 * D_MODEL * HIDDEN,
                             "step_mfma_exec_frac": value * 6 * D_MODEL * HIDDEN / 1e12 / BF16_DENSE_PEAK_TFLOPS})
        out = {
            "metric": f"activations/sec through SAE train step (d={D_MODEL}->{HIDDEN}, " + ("relu+l1)" if args.relu else f"k={TOPK})"),
            "value": value, "unit": "activations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "windows_ms_per_step": [w / args.steps * 1e3 for w in windows],
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": (f"ReLUSAE {D_MODEL}->{HIDDEN} (sparsity_weight 0.01{', fp8 forward GEMMs' if args.fp8 else ''}) train step, informational" if args.relu else
                                    f"TopKSAE {D_MODEL}->{HIDDEN} k={TOPK} train step, informational" if args.dims else
                                    "BASELINE.json configs[1]: TopKSAE 384->3072 k=32 train step") + ", synthetic "
                                   "activations resident in the HBM ring buffer" + ("" if world == 1 else
                                   f" (configs[2]: DDP x{world}, one RCCL all-reduce per step of the gradients, the dead-feature "
                                   f"indicators and the step's metric scalars on a {trainer.grad_exchange} wire, issued in stream order)"),
                       "batch_per_gpu": B, "global_batch": world * B, "ring_rows_per_gpu": args.ring_rows,
                       "lr": 1e-4, "clip": 1.0, "parallelism": f"dp{world}",
                       **({"grad_exchange": trainer.grad_exchange} if world > 1 else {})},
            **({"rehearsal": True} if rehearse else {}),
            "roofline": roof,
            "roofline_wgrad2": roof_wgrad,
            "probe_kernel_us": {k: round(v[1] / max(v[0], 1) * 1e3, 2) for k, v in probe.items()},
            "untimed_steps_before_windows": {"warmup": args.warmup, "settle": n_settle, "probe": n_probe},
            "step_dense_equiv_frac": value * f_dense / 1e12 / BF16_DENSE_PEAK_TFLOPS,
            "final_loss": last.loss if last is not None else None,
        }
        if args.profile_all:
            out["kernel_ms_per_step"] = {k: v[1] / max(v[0], 1) for k, v in prof.items()}
        if not args.relu:
            # selective strip stores of the encoder GEMM (include/wsae.h): rows that had to recompute strips over the whole
            # run (warm-up, timed windows, probe) and the last batch's smallest row threshold
            import ctypes
            rows_c, tmin_c, s_c = ctypes.c_int64(0), ctypes.c_float(0.0), ctypes.c_float(0.0)
            N.check(N.lib().wsae_ctx_strip_stats(handle, ctypes.addressof(rows_c), ctypes.addressof(tmin_c), ctypes.addressof(s_c)),
                    "wsae_ctx_strip_stats")
            out["strip_predict"] = {"enabled": os.environ.get("WSAE_STRIP_PREDICT", "1") != "0", "refilled_rows": rows_c.value,
                                    "last_min_threshold": tmin_c.value, "margin": s_c.value}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(B)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
