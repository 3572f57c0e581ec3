#!/usr/bin/env python3
"""Headline benchmark: point-clouds/sec, forward + backward (+ Adam step and gradient all-reduce),
20 000-point GraspNet-like scenes, B = 4 clouds per GPU — BASELINE.json configs[3] (the configuration
the metric "fwd+bwd ... 1/2/4/8 MI355X" is quoted on; it fits one GPU).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One JSON line on rank 0.  A step = GraspBalance forward (training mode, label matching included) ->
loss -> backward -> flat-bucket RCCL all-reduce -> Adam -> LR step, on a batch already resident in HBM.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

BATCH_PER_GPU = 4
NUM_POINT = 20000
HBM_PEAK_GBS = 8000.0
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD


def fps_algorithmic_bytes(b, n, m):
    """SURVEY.md §8d streaming model: 12 B xyz + 4 B read + 4 B write of the running min-distance per
    point per iteration, plus the index output."""
    return b * (20.0 * n * (m - 1) + 4.0 * m)


def pmc_traffic(kernel):
    """HBM bytes per launch from the committed PMC passes (profiles/r01_pmc_traffic.json: separate
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this same command, gfx950 corrections applied);
    None when the file is absent.  Counters cannot be collected from inside this process."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
            return round(json.load(f)[kernel]["hbm_bytes_per_launch"], 1)
    except Exception:
        return None


def cpu_baseline(num_threads):
    """The same train step on the host cores, with the CPU oracle (oracle/graspbal_oracle.c) standing
    in for the HIP extension and torch-CPU for the MLPs: a bounded sample of ONE step on ONE cloud."""
    from tests import cpu_backend
    from graspbalance_amd.synthetic import make_training_batch
    from graspbalance_amd.train import Trainer
    from graspbalance_amd import knn_modules, pointnet2_utils
    from graspbalance_amd.modified_net_tools import group, subsample, upsampling
    saved = [(pointnet2_utils, "_ext", pointnet2_utils._ext), (group, "pointnet2_cuda", group.pointnet2_cuda),
             (subsample, "pointnet2_cuda", subsample.pointnet2_cuda),
             (upsampling, "pointnet2_cuda", upsampling.pointnet2_cuda), (knn_modules, "knn", knn_modules.knn)]
    torch.set_num_threads(num_threads)
    try:
        cpu_backend.install()
        trainer = Trainer("cpu")
        trainer.net.grasp_generator.fused_cylinder = False  # the reference issues 16 separate queries
        batch = make_training_batch([0], NUM_POINT, device="cpu")
        t0 = time.time()
        loss = trainer.train_step(batch)
        dt = time.time() - t0
        assert bool(torch.isfinite(loss))
    finally:
        for mod, name, val in saved:
            setattr(mod, name, val)
    return {"value": 1.0 / dt, "unit": "point-clouds/s", "cores": num_threads, "kind": "port",
            "sample": "1 train step (fwd+bwd+Adam) on 1 cloud of %d points, oracle C geometry (OpenMP) + torch CPU MLPs, %.1f s"
                      % (NUM_POINT, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d needs WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no GPU visible)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # GB_FORCE_DIST=1 runs the RCCL path (broadcast, flat-bucket all-reduce, barriers) even with one rank
    use_dist = world > 1 or os.environ.get("GB_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from graspbalance_amd import _lib
    from graspbalance_amd.synthetic import make_training_batch
    from graspbalance_amd.train import Trainer
    _lib.lib()  # fail loudly if the HIP library is missing

    trainer = Trainer(device, distributed=use_dist)
    seeds = [1000 * rank + i for i in range(BATCH_PER_GPU)]
    batch = make_training_batch(seeds, NUM_POINT, device=device)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        trainer.train_step(batch)
    barrier()
    # ~240 timed launches per step, two events each: created before the timed region, recorded inside it
    timer = _lib.KernelTimer(["gb_fps", "gb_gemm_fwd", "gb_gemm_fwd_w", "gb_gemm_dgrad", "gb_gemm_dgrad_first",
                              "gb_gemm_wgrad"], reserve=min(2 * 260 * args.steps, 20000))
    barrier()
    with timer as kt:
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = trainer.train_step(batch)
        barrier()
        elapsed = time.perf_counter() - t0
    assert bool(torch.isfinite(loss)), "training diverged"
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        clouds = world * BATCH_PER_GPU * args.steps
        # dominant kernels of the step by total time: the two fp32 MFMA GEMM kernels of the fused SharedMLP
        # path (gemm_cl_kernel: LDS-tiled; gemm_rs_kernel: row-streaming).  "roofline" is whichever has the
        # larger share, the other goes to "roofline_gemm2".  Per launch: algorithmic FLOP = 2*P*K*N;
        # achieved = sum FLOP / sum launch durations (HIP events on the launch stream, in the timed region).
        ev_bias = _lib.event_pair_overhead_ms(device)  # what an empty event bracket reads; removed from every launch

        def gemm_roofline(kernel, what):
            ev = [(max(a.elapsed_time(b) - ev_bias, 1e-4), m["flop"]) for n in ("gb_gemm_fwd", "gb_gemm_fwd_w", "gb_gemm_dgrad", "gb_gemm_dgrad_first", "gb_gemm_wgrad")
                  for a, b, m in kt.events[n] if m["kernel"] == kernel]
            if not ev:
                return None
            ms = sum(t for t, _ in ev)
            flop = sum(f for _, f in ev)
            achieved = flop / (ms * 1e-3) / 1e12
            return {"kernel": "%s (v_mfma_f32_32x32x2_f32; %s)" % (kernel, what), "bound": "mfma",
                    "achieved": round(achieved, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / MFMA_F32_PEAK_TFLOPS, 4), "traffic": pmc_traffic(kernel),
                    "launch_ms": round(ms / len(ev), 4), "launches": len(ev),
                    "ms_per_step": round(ms / args.steps, 3), "gflop_per_launch": round(flop / len(ev) / 1e9, 3)}

        rl_cl = gemm_roofline("gemm_cl_kernel", "LDS-tiled: split-K wgrad, small / unaligned fwd and dgrad")
        rl_rs = gemm_roofline("gemm_rs_kernel", "row-streaming: tall fwd+BN-stats and dgrad+BN-backward sums")
        both = sorted([r for r in (rl_cl, rl_rs) if r], key=lambda r: -r["ms_per_step"])
        roofline = both[0] if both else None
        roofline_second = both[1] if len(both) > 1 else None
        # largest single launch of the step: the first-level FPS (HBM class, streaming-model bytes)
        roofline_fps = None
        big = [(a.elapsed_time(b) - ev_bias, m) for a, b, m in kt.events["gb_fps"] if m["n"] == NUM_POINT]
        if big:
            mean_ms = sum(x for x, _ in big) / len(big)
            meta = big[0][1]
            achieved = fps_algorithmic_bytes(meta["b"], meta["n"], meta["m"]) / (mean_ms * 1e-3) / 1e9
            pruned = _lib._fps_prune and _lib.FPS_PRUNE_MIN_N <= meta["n"] <= _lib.FPS_PRUNE_MAX_N
            kname = "fps_pruned_kernel" if pruned else "fps_reg_kernel<1024, 20>"
            order = "cell-order counting sort" if _lib._fps_cell_order else "Morton keys + sort"
            what = order + " + fps_pruned_kernel<1024,20>" if pruned else "fps_reg_kernel<1024,20>"
            roofline_fps = {"kernel": "%s (furthest_point_sampling %d->%d, b=%d)" % (what, meta["n"], meta["m"], meta["b"]),
                            "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(kname),
                            "launch_ms": round(mean_ms, 4), "launches": len(big)}
        out = {
            "metric": "point-clouds/sec fwd+bwd, 20k-pt GraspNet scene",
            "value": round(clouds / elapsed, 3), "unit": "point-clouds/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic (make_scene clouds + seeded uniform grasp labels; no dataset available)",
            "config": {"workload": "configs[3]: GraspBalance train step fwd+bwd+Adam, B=%d/GPU, N=%d points, "
                                   "8 objects x 300 grasp points x 300 views labels" % (BATCH_PER_GPU, NUM_POINT),
                       "global_batch": world * BATCH_PER_GPU, "parallelism": "dp%d" % world},
            "roofline": roofline,
            "roofline_gemm2": roofline_second,
            "event_bias_us": round(ev_bias * 1e3, 2),
            "roofline_fps": roofline_fps,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(min(os.cpu_count() or 1, 32))
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
