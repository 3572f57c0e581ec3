#!/usr/bin/env python3
"""Headline benchmark: point-clouds/sec, forward + backward (+ Adam step and gradient all-reduce),
20 000-point GraspNet-like scenes, B = 4 clouds per GPU — BASELINE.json configs[3] (the configuration
the metric "fwd+bwd ... 1/2/4/8 MI355X" is quoted on; it fits one GPU).

    python bench.py --gpus N --steps K --warmup W          # N > 1: spawns its own N ranks (torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One JSON line on rank 0.  A step = GraspBalance forward (training mode, label matching included) ->
loss -> backward -> flat-bucket RCCL all-reduce -> Adam -> LR step, on a batch already resident in HBM.

Execution: the step is captured once in a HIP graph (train.Trainer(graph=True)) and the K timed steps replay it - one
launch call per step instead of ~1000 (GB_GRAPH=0: every step enqueued launch by launch, as in rounds 1-3).  A replayed
graph cannot be bracketed launch by launch, so the per-kernel roofline figures come from a few EAGER steps of the same
trainer right after the timed region (the same kernels on the same shapes: `roofline.measured_on` says so); the
rocprofv3 kernel trace of this command (profiles/) sees the replayed launches themselves.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # before anything initialises HIP: dmabuf IPC for RCCL

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BATCH_PER_GPU = 4
NUM_POINT = 20000
HBM_PEAK_GBS = 8000.0
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
MFMA_BF16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA (16x the fp32 rate)
# vector ALU: 256 CUs x 4 SIMDs x 32 lanes/clk x 2.4 GHz = 78.6e12 lane-instructions/s (the 157.3 TFLOP/s vector
# figure counts an FMA as two; the geometry kernels are built -ffp-contract=off, one rounding per operation)
VALU_PEAK_TLANEOPS = 256 * 4 * 32 * 2.4e9 / 1e12
# lane-instructions per scanned (centre, candidate) pair, from the kernels' arithmetic:
BALL_OPS_PER_PAIR = 9        # 3 sub, 3 mul, 2 add, 1 compare
CYL16_OPS_PER_PAIR = 30      # 3 sub, 9 mul + 6 add (rotate), 2 mul + 1 add (radial), 4 + 4 + 1 compares (4 radii x 4 hmax, hmin)


def _pin_rank_to_cores(local_rank, local_world):
    """Give this rank its own host cores BEFORE anything initialises the GPU: the cores of its GPU's NUMA node (sysfs:
    amdgpu PCI functions in bus order = HIP's device order on this image) shared evenly among the ranks on that node, or -
    where sysfs does not say - an even slice of the cores this process may use.  Eight Python ranks left to the
    scheduler migrate across sockets and contend for the cores that enqueue their steps.  Returns a short description
    for the JSON line (never raises: pinning is an optimisation)."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
        if os.environ.get("GB_BENCH_PIN", "1") == "0" or local_world <= 1 or len(allowed) < 2 * local_world:
            return {"cores": len(allowed), "numa_aware": False, "pinned": False}
        nodes = []
        base = "/sys/bus/pci/drivers/amdgpu"
        if os.path.isdir(base):
            for dev in sorted(d for d in os.listdir(base) if ":" in d):
                try:
                    with open(os.path.join(base, dev, "numa_node")) as f:
                        nodes.append(int(f.read().strip()))
                except (OSError, ValueError):
                    nodes.append(-1)
        mine = None
        if len(nodes) >= local_world and nodes[local_rank] >= 0:
            node = nodes[local_rank]
            try:
                with open("/sys/devices/system/node/node%d/cpulist" % node) as f:
                    cpus = set()
                    for part in f.read().strip().split(","):
                        lo, _, hi = part.partition("-")
                        cpus.update(range(int(lo), int(hi or lo) + 1))
                cpus = sorted(cpus & set(allowed))
                peers = [r for r in range(local_world) if nodes[r] == node]
                share = len(cpus) // len(peers)
                if share >= 1:
                    k = peers.index(local_rank)
                    mine = cpus[k * share:(k + 1) * share]
            except (OSError, ValueError):
                mine = None
        numa = mine is not None
        if mine is None:
            share = len(allowed) // local_world
            mine = allowed[local_rank * share:(local_rank + 1) * share]
        os.sched_setaffinity(0, mine)
        return {"cores": len(mine), "numa_aware": numa, "pinned": True}
    except Exception as e:  # noqa: BLE001
        return {"cores": None, "numa_aware": False, "pinned": False, "error": repr(e)[:80]}


def _spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start N fresh worker processes (one per GPU) through
    torch.distributed.run BEFORE this process touches the GPU, relay rank 0's JSON line, exit with the job's code."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    return subprocess.call(cmd, env=env)


def fps_algorithmic_bytes(b, n, m):
    """SURVEY.md section 8d streaming model: 12 B xyz + 4 B read + 4 B write of the running min-distance per
    point per iteration, plus the index output."""
    return b * (20.0 * n * (m - 1) + 4.0 * m)


KERNEL_SOURCES = {"gemm_rs_kernel": ("gemm_rs.hip", "gemm_rs.h"), "gemm_ring_kernel": ("gemm_ring.hip", "gemm_ring.h"),
                  "gemm_ring_group_kernel": ("gemm_ring.hip", "gemm_ring.h"),
                  "gemm_cl_kernel": ("gemm_cl.hip",), "wgrad_direct_kernel": ("gemm_wg.hip", "gemm_wg.h"), "fps_rows_kernel": ("fps.hip",), "fps_pruned_kernel": ("fps.hip",),
                  "fps_reg_kernel<1024, 20>": ("fps.hip",)}
PMC_FILE = "r06_pmc_traffic.json"


def git_blob_hash(path):
    """The hash `git hash-object` gives the file (sha1 of "blob <size>\\0" + content): ties a measurement to the source."""
    import hashlib
    with open(path, "rb") as f:
        data = f.read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def kernel_source_hashes(kernel):
    src = os.path.join(ROOT, "graspbalance_amd", "csrc")
    return {name: git_blob_hash(os.path.join(src, name)) for name in KERNEL_SOURCES.get(kernel, ())}


def pmc_traffic(kernel, with_source=False):
    """HBM bytes per launch from the committed PMC passes (profiles/r05_pmc_traffic.json: separate rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE runs of this same command, gfx950 corrections applied).  Counters cannot be collected from
    inside this process, so this is a figure from a file - and it only counts while the kernel's SOURCE is the one the
    counters were taken on: the file records the git blob hashes of the kernel's .hip / .h files (tools/pmc_summary.py),
    and a kernel whose source has changed since gets `traffic: null`."""
    source = {"file": "profiles/" + PMC_FILE, "kernel_sources": None, "valid": False}
    try:
        with open(os.path.join(ROOT, "profiles", PMC_FILE)) as f:
            ent = json.load(f)[kernel]
        want = ent.get("sources")
        have = kernel_source_hashes(kernel)
        source["kernel_sources"] = want
        source["valid"] = bool(want) and want == have
        val = round(ent["hbm_bytes_per_launch"], 1) if source["valid"] else None
    except Exception:
        val = None
    return (val, source) if with_source else val


def _config_leg(argv):
    """One of the other BASELINE configurations, run by THIS script in a child process (a fresh allocator and trainer; the
    parent never execs - the child is a child) and condensed to what the headline line carries for it."""
    env = dict(os.environ)
    env["GB_BENCH_CHANGING"] = "0"
    env["GB_BENCH_GRAPH_GUARD"] = "0"
    try:
        res = subprocess.run([sys.executable, os.path.abspath(__file__)] + argv, capture_output=True, text=True, timeout=300, env=env)
        line = [l for l in res.stdout.splitlines() if l.startswith("{")]
        if res.returncode != 0 or not line:
            return {"error": (res.stderr or res.stdout)[-400:]}
        d = json.loads(line[-1])
    except Exception as e:   # the headline must not die of a side leg
        return {"error": repr(e)[:400]}
    keep = ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "dtype", "config", "roofline", "roofline_fps",
            "ms_per_step_next_batch_announced", "ms_per_step_no_prefetch", "max_memory_allocated_gb")
    return {k: d[k] for k in keep if k in d}


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2]


def cpu_baseline(num_threads, device, repeats=5):
    """SURVEY.md section 8d leg (ii), the graded ``cpu_baseline``: the SAME train step (forward incl. label matching,
    loss, backward, Adam) on the host cores, with the C oracle (oracle/graspbal_oracle.c, OpenMP) standing in for the
    HIP extension and torch-CPU for the MLPs.  Bounded sample: ONE cloud per step; 1 warm-up + median of `repeats`
    steps (time.perf_counter).  Before the number counts, the first step's loss is cross-checked against the HIP
    path's first step from the same seed on the same cloud."""
    import torch
    from tests import cpu_backend
    from graspbalance_amd.synthetic import make_training_batch
    from graspbalance_amd.train import Trainer
    from graspbalance_amd import knn_modules, pointnet2_utils
    from graspbalance_amd.modified_net_tools import group, subsample, upsampling
    saved = [(pointnet2_utils, "_ext", pointnet2_utils._ext), (group, "pointnet2_cuda", group.pointnet2_cuda),
             (subsample, "pointnet2_cuda", subsample.pointnet2_cuda),
             (upsampling, "pointnet2_cuda", upsampling.pointnet2_cuda), (knn_modules, "knn", knn_modules.knn)]
    torch.set_num_threads(num_threads)
    gpu_trainer = Trainer(device, graph=False)   # one step: nothing to replay
    probe = make_training_batch([0], NUM_POINT, device=device)['point_clouds']
    keys = ('sa1_features', 'fp2_features', 'objectness_score', 'view_score', 'grasp_score_pred', 'grasp_width_pred')

    def eval_forward(tr, clouds):   # same seed -> same initial weights on both sides; eval mode: no batch statistics
        tr.net.eval()
        tr.net.is_training = tr.net.view_estimator.is_training = tr.net.grasp_generator.is_training = False
        try:
            with torch.no_grad():
                out = tr.net({'point_clouds': clouds})
            return {k: out[k].detach().float().cpu() for k in keys}, out['grasp_top_view_inds'].cpu()
        finally:
            tr.net.train()
            tr.net.is_training = tr.net.view_estimator.is_training = tr.net.grasp_generator.is_training = True
    gpu_eval, gpu_views = eval_forward(gpu_trainer, probe)
    gpu_loss = float(gpu_trainer.train_step(make_training_batch([0], NUM_POINT, device=device)).detach())
    try:
        cpu_backend.install()
        trainer = Trainer("cpu")
        trainer.net.grasp_generator.fused_cylinder = False  # the reference issues 16 separate queries
        # tight cross-check before anything is timed: the eval-mode forward of the SAME network on the same cloud, HIP
        # path vs this CPU path, tensor by tensor (the train-step loss below sits behind batch-statistic BatchNorms at
        # B = 1 and top-view arg-max flips: it agrees to 3e-3 only, which would not catch a wrong term)
        cpu_eval, cpu_views = eval_forward(trainer, probe.cpu())
        same = (gpu_views == cpu_views).all(dim=1, keepdim=True)        # seeds whose top view agrees
        agree_eval = {}
        for k in keys:
            a, b = gpu_eval[k].double(), cpu_eval[k].double()
            agree_eval[k] = float((a - b).norm() / (b.norm() + 1e-30))
        flips = int((gpu_views != cpu_views).sum())
        stage1 = max(agree_eval[k] for k in ('sa1_features', 'fp2_features', 'objectness_score', 'view_score'))
        assert stage1 < 1e-4, "CPU and HIP eval forwards disagree: %r" % agree_eval
        assert flips > 0 or max(agree_eval.values()) < 1e-4, "CPU and HIP eval forwards disagree: %r" % agree_eval
        # the label generator is device-specific: build on the GPU's generator, move to the host
        batch = {k: ([[t.cpu() for t in per] for per in v] if isinstance(v, list) else v.cpu())
                 for k, v in make_training_batch([0], NUM_POINT, device=device).items()}
        times, first = [], None
        for i in range(repeats + 1):
            t0 = time.perf_counter()
            loss = trainer.train_step(batch)
            dt = time.perf_counter() - t0
            if first is None:
                first = float(loss.detach())
            if i:
                times.append(dt)
            assert bool(torch.isfinite(loss))
    finally:
        for mod, name, val in saved:
            setattr(mod, name, val)
    agree = abs(first - gpu_loss) / max(1.0, abs(gpu_loss))
    assert agree < 5e-3, "CPU and HIP first-step losses disagree: %r vs %r" % (first, gpu_loss)
    med = _median(times)
    return {"value": round(1.0 / med, 4), "unit": "point-clouds/s", "cores": num_threads, "kind": "port",
            "sample": "train step (fwd+bwd+Adam) on 1 cloud of %d points: 1 warm-up + median of %d steps (%.2f s each), "
                      "oracle C geometry (OpenMP) + torch CPU MLPs" % (NUM_POINT, repeats, med),
            "cpu_model": _cpu_model(), "os_cpu_count": os.cpu_count(), "torch_threads": torch.get_num_threads(),
            "first_step_loss": {"cpu": round(first, 6), "hip": round(gpu_loss, 6)},
            "eval_forward_rel_l2_hip_vs_cpu": {k: float("%.2e" % v) for k, v in agree_eval.items()},
            "eval_top_view_flips": flips}


def cpu_baseline_dense(num_threads, device, repeats=5):
    """SURVEY.md section 8d leg (i): the dense-torch restatement of the reference's CPU fallback algorithm
    (oracle/dense_torch.py: distance matrix + sort, whole-tensor FPS) on BASELINE configs[1] - one SA layer
    (npoint 1024, r 0.04, ns 32, MLP [3,64,128]) forward on a 20 000-point cloud - next to the HIP path on the same
    cloud.  Outputs are cross-checked (FPS and ball-query indices identical, features to 1e-4) before timing:
    1 warm-up + median of `repeats`."""
    import torch
    from graspbalance_amd import pointnet2_modules as pm
    from graspbalance_amd.scene import make_batch
    from oracle import dense_torch as dt
    torch.set_num_threads(num_threads)
    torch.manual_seed(11)
    sa = pm.PointnetSAModuleVotes(npoint=1024, radius=0.04, nsample=32, mlp=[0, 64, 128], use_xyz=True,
                                  normalize_xyz=True).train()
    weights = [(l.conv.weight.detach().view(l.conv.weight.shape[0], -1).clone(), l.bn.bn.weight.detach().clone(),
                l.bn.bn.bias.detach().clone()) for l in sa.mlp_module.children()]
    cloud = torch.from_numpy(make_batch([0], NUM_POINT))
    sa = sa.to(device)
    xyz = cloud.to(device)
    cpu_t, gpu_t = [], []
    for i in range(repeats + 1):
        t0 = time.perf_counter()
        inds, idx, feats = dt.sa_layer_forward(cloud, 1024, 0.04, 32, weights)
        if i:
            cpu_t.append(time.perf_counter() - t0)
    with torch.no_grad():
        for i in range(repeats + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            new_xyz, got, got_inds = sa(xyz)
            torch.cuda.synchronize()
            if i:
                gpu_t.append(time.perf_counter() - t0)
    from graspbalance_amd import pointnet2_utils as pu
    # the module samples with the CUDA kernel's tree tie-break, the fallback algorithm with the lowest index: they
    # differ only between exact duplicates of a point, so the sampled COORDINATES must agree - and the HIP kernel in
    # lowest-index mode must reproduce the dense indices themselves
    want_xyz = torch.gather(cloud, 1, inds.long().unsqueeze(-1).expand(-1, -1, 3))
    assert torch.equal(new_xyz.cpu(), want_xyz), "FPS samples differ between the HIP path and the dense restatement"
    from graspbalance_amd import _lib
    low = torch.empty((1, 1024), dtype=torch.int32, device=device)
    tmp = torch.full((1, NUM_POINT), 1e10, dtype=torch.float32, device=device)
    _lib.check(_lib.fps(xyz, tmp, low, 1, NUM_POINT, 1024, _lib.FPS_SKIP_NEAR_ORIGIN | _lib.FPS_TIE_LOWEST,
                        _lib.current_stream(device)), "fps")
    assert torch.equal(low.cpu(), inds), "FPS indices (lowest-index ties) differ from the dense restatement"
    assert torch.equal(pu.ball_query(0.04, 32, xyz, new_xyz).cpu(), idx), "ball-query indices differ"
    err = float((got.cpu() - feats).norm() / feats.norm())
    assert err < 1e-4, err
    c, g = _median(cpu_t), _median(gpu_t)
    return {"workload": "configs[1]: one SA layer (npoint 1024, r 0.04, ns 32, MLP [3,64,128]) forward, 1 cloud of %d points"
                        % NUM_POINT, "cpu_value": round(1.0 / c, 3), "hip_value": round(1.0 / g, 1),
            "unit": "point-clouds/s", "ratio": round(c / g, 1), "cores": num_threads, "kind": "port (dense torch)",
            "feature_rel_err": err, "sample": "1 warm-up + median of %d (CPU %.3f s, HIP %.2f ms per cloud)" % (repeats, c, g * 1e3)}


def infer_main(args):
    """BASELINE configs[2]: full GraspBalance forward (eval mode, running statistics) + pred_decode on B = 4 clouds of
    20 000 points per GPU, clouds resident in HBM.  Nothing hides the first-level furthest-point sampling here (no next
    batch is announced): its launch is timed with HIP events inside the timed region and priced against the streaming
    model like the train line's.  Ranks are independent replicas (inference has no collective)."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no GPU visible)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    from graspbalance_amd import _lib
    from graspbalance_amd.graspbalance import GraspBalance
    from graspbalance_amd.predict import Predictor
    from graspbalance_amd.scene import make_batch
    _lib.lib()
    torch.manual_seed(1234)
    predictor = Predictor(GraspBalance(is_training=False), device)
    clouds = torch.from_numpy(make_batch([1000 * rank + i for i in range(BATCH_PER_GPU)], NUM_POINT)).to(device)
    batch = {'point_clouds': clouds}

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def step():
        return predictor(batch)   # no next batch announced: the sampling is on the critical path (one request's latency)
    for _ in range(args.warmup):
        step()
    barrier()
    timer = _lib.KernelTimer(["gb_fps", "gb_ball_query"], reserve=64 * args.steps)
    with timer as kt:
        t0 = time.perf_counter()
        for _ in range(args.steps):
            grasps = step()
        barrier()
        elapsed = time.perf_counter() - t0
    # the same K steps as a loop that holds its next batch (a dataset sweep, a request queue): the next batch's first-level
    # sampling runs on a side stream under this forward (predict.Predictor, like Trainer.train_step(batch, next_batch=))
    predictor(batch, next_batch=batch)
    barrier()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        grasps_p = predictor(batch, next_batch=batch)
    barrier()
    pipelined = time.perf_counter() - t1
    predictor(batch)   # consume the last announced sampling
    t = torch.tensor([elapsed, pipelined], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed, pipelined = float(t[0].item()), float(t[1].item())
    if rank == 0:
        assert all(torch.equal(a, b) for a, b in zip(grasps, grasps_p))   # same grasps either way
        ev_bias = _lib.event_pair_overhead_ms(device)
        big = [(a.elapsed_time(b) - ev_bias, m) for a, b, m in kt.events["gb_fps"] if m["n"] == NUM_POINT]
        fps_ms = sum(x for x, _ in big) / len(big)
        meta = big[0][1]
        fps_bytes = fps_algorithmic_bytes(meta["b"], meta["n"], meta["m"])
        ball = [(a.elapsed_time(b) - ev_bias, m) for a, b, m in kt.events["gb_ball_query"] if m["n"] == NUM_POINT]
        ball_ms = sum(x for x, _ in ball) / len(ball)
        ach = fps_bytes / (fps_ms * 1e-3) / 1e9
        out = {"metric": "point-clouds/sec forward (inference), 20k-pt GraspNet scene", "value": round(world * BATCH_PER_GPU * args.steps / elapsed, 3),
               "unit": "point-clouds/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f32", "data": "synthetic (make_scene clouds, default-initialised weights)",
               "config": {"workload": "configs[2]: GraspBalance eval forward + pred_decode, B=%d/GPU, N=%d points"
                                      % (BATCH_PER_GPU, NUM_POINT), "global_batch": world * BATCH_PER_GPU,
                          "parallelism": "replicas x%d" % world},
               "roofline": {"kernel": "row-order counting sort + fps_rows_kernel (furthest_point_sampling %d->%d, b=%d), "
                                      "on the critical path" % (meta["n"], meta["m"], meta["b"]),
                            "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": pmc_traffic("fps_rows_kernel"),
                            "launch_ms": round(fps_ms, 4), "launches": len(big),
                            "share_of_step": round(fps_ms / (elapsed / args.steps * 1e3), 3)},
               "first_level_ball_query_ms": round(ball_ms, 4),
               "max_memory_allocated_gb": round(torch.cuda.max_memory_allocated(device) / 2 ** 30, 2),
               "ms_per_step_next_batch_announced": round(pipelined / args.steps * 1e3, 3),
               "value_next_batch_announced": round(world * BATCH_PER_GPU * args.steps / pipelined, 3)}
        assert len(grasps) == BATCH_PER_GPU and all(g.shape[1] == 17 for g in grasps)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="default train line only: skip the short configs[2] / configs[4] legs printed under `configs`")
    ap.add_argument("--config", choices=["train", "stress", "infer"], default="train",
                    help="train = BASELINE configs[3] (the headline line); stress = configs[4]: B=8/GPU, N=50000, bf16 MLP; "
                         "infer = configs[2]: eval forward + pred_decode, B=4, N=20000 (first-level FPS on the critical path)")
    args = ap.parse_args()
    global BATCH_PER_GPU, NUM_POINT
    stress = args.config == "stress"
    if stress:
        BATCH_PER_GPU, NUM_POINT = 8, 50000

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(_spawn_ranks(args.gpus, sys.argv[1:]))  # nothing has touched the GPU in this process
    if args.config == "infer":
        return infer_main(args)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d was launched with WORLD_SIZE=%d" % (args.gpus, world))
    affinity = _pin_rank_to_cores(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)))   # before any GPU call
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no GPU visible)")
    # GB_REHEARSE_ON_ONE_GPU=1 (not a measurement: the line says so): every rank uses device 0 and the collectives run
    # on gloo, so that the N-rank code path - launcher, per-rank batches, bucket all-reduce, max over ranks, rank 0's
    # line - can be exercised on a one-GPU box.  RCCL refuses two ranks on one device.
    rehearsal = os.environ.get("GB_REHEARSE_ON_ONE_GPU") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # GB_FORCE_DIST=1 runs the RCCL path (broadcast, flat-bucket all-reduce, barriers) even with one rank
    use_dist = world > 1 or os.environ.get("GB_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from graspbalance_amd import _lib
    from graspbalance_amd.synthetic import make_training_batch
    from graspbalance_amd.train import Trainer
    _lib.lib()  # fail loudly if the HIP library is missing

    # GB_BENCH_PRECISION is a probe switch (what would the step cost with cheaper contractions); the default line is f32
    prec = os.environ.get("GB_BENCH_PRECISION", "bf16" if stress else "f32")
    trainer = Trainer(device, distributed=use_dist, time_collectives=use_dist, mlp_precision=prec)
    seeds = [1000 * rank + i for i in range(BATCH_PER_GPU)]
    batch = make_training_batch(seeds, NUM_POINT, device=device)
    # graph execution: the trainer's static input buffers ARE the resident batch (no staging copy in the steps)
    batch = trainer.resident(batch)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    # the loop holds its next batch (here: the same resident batch), so every step announces it: the first-level
    # FPS of step t+1 runs on a side stream under step t - once per step, inside the timed region, never cached
    # (graph execution: the capture happens in the first warm-up step)
    for _ in range(max(args.warmup, 1 if trainer.graph else 0)):
        trainer.train_step(batch, next_batch=batch)
    barrier()
    # Guard, outside the timed region: graph replay is the product's default because it is faster, which holds on every
    # configuration measured (one GPU, one-rank RCCL group) - but that is hardware / runtime behaviour, not a law: with
    # several processes sharing ONE GPU (the rehearsal mode) every cross-stream dependency costs a scheduling quantum and a
    # replayed step takes seconds.  Two steps each way - the launch-by-launch reference on a trainer of its own that never
    # captured anything; a job whose replay is more than twice as slow runs launch by launch and says so in `execution`.
    # All ranks take the same decision.
    graph_fallback = None
    t_eager_clean = None
    if trainer.graph and os.environ.get("GB_BENCH_GRAPH_GUARD", "1") != "0":
        def per_step(tr, b, n=2):
            barrier()
            t = time.perf_counter()
            for _ in range(n):
                tr.train_step(b, next_batch=b)
            torch.cuda.synchronize()
            return (time.perf_counter() - t) / n * 1e3
        probe = Trainer(device, distributed=use_dist, mlp_precision=prec, graph=False)
        per_step(probe, batch, 2)
        t_eager = per_step(probe, batch)
        t_eager_clean = t_eager      # launch-by-launch execution as shipped (graph=False), no instrumentation: goes into the line
        del probe
        t_graph = per_step(trainer, batch)
        flag = torch.tensor([1.0 if t_graph > 2.0 * t_eager else 0.0], device=device)
        if use_dist:
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if float(flag.item()) > 0:
            graph_fallback = "graph replay measured %.1f ms per step against %.1f ms launch by launch on this job" % (t_graph, t_eager)
            trainer = Trainer(device, distributed=use_dist, time_collectives=use_dist, mlp_precision=prec, graph=False)
            for _ in range(max(args.warmup, 2)):
                trainer.train_step(batch, next_batch=batch)
        barrier()
    if use_dist:
        trainer.grads.exposed_ms()  # drop the warm-up samples
    from graspbalance_amd import fused_mlp as _fm
    _fm.SYNC_WAIT[0] = 0.0
    replays0 = trainer.graph_replays
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = trainer.train_step(batch, next_batch=batch)
    # how long the host WORKED to enqueue the K steps (a step waits for the GPU nowhere any more: the crop row counts
    # stay on the device).  Well below `elapsed`: the step is GPU-bound.  Close to it: the step is host-bound and the
    # line says more about the box's CPU (and its other tenants) than about the kernels.
    host_enqueue = time.perf_counter() - t0 - _fm.SYNC_WAIT[0]
    host_waited = _fm.SYNC_WAIT[0]   # (graph loop: the host runs at most three steps ahead of the GPU and waits there)
    barrier()
    elapsed = time.perf_counter() - t0
    assert bool(torch.isfinite(loss)), "training diverged"
    assert not trainer.graph or trainer.graph_replays - replays0 == args.steps, "a timed step did not replay the graph"
    # ---- per-kernel durations: the same step enqueued launch by launch, every launch of interest bracketed with HIP
    # events on its stream (the brackets cost ~2 us per launch, and a replayed graph cannot be bracketed at all): AFTER the
    # timed region, same trainer, same resident batch, same kernels
    gemm_names = ["gb_gemm_fwd", "gb_gemm_fwd_w", "gb_gemm_fwd_pool", "gb_gemm_fwd_gen3", "gb_gemm_dgrad", "gb_gemm_dgrad_first",
                  "gb_gemm_dgrad_first_gen3", "gb_gemm_wgrad", "gb_gemm_wgrad_gen3", "gb_gemm_dgrad_wgrad", "gb_gemm_wgrad_group"]
    # GB_BENCH_TIMED_ONLY=1 (profiling runs: tools/refresh_profiles.sh): nothing but warm-up and the timed replays, so that
    # a kernel trace's last steps ARE the timed ones
    timed_only = os.environ.get("GB_BENCH_TIMED_ONLY") == "1"
    if timed_only:
        if rank == 0:
            print(json.dumps({"metric": "point-clouds/sec fwd+bwd (timed steps only: profiling run)", "value": round(world * BATCH_PER_GPU * args.steps / elapsed, 3),
                              "ms_per_step": round(elapsed / args.steps * 1e3, 3), "steps": args.steps,
                              "host_enqueue_ms_per_step": round(host_enqueue / args.steps * 1e3, 3)}))
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        return
    sampled = max(2, min(5, args.steps // 4))
    timer = _lib.KernelTimer(gemm_names + ["gb_fps", "gb_ball_query", "gb_cylinder_query_multi"],
                             reserve=min(2 * 300 * (sampled + 2), 20000))
    trainer.train_step_eager(batch, next_batch=batch)   # (announces the next sampling for the first bracketed step)
    barrier()
    with timer as kt:
        kt.sample(True)
        t1 = time.perf_counter()
        for _ in range(sampled):
            trainer.train_step_eager(batch, next_batch=batch)
        eager_host = time.perf_counter() - t1
        torch.cuda.synchronize()
        eager_ms = (time.perf_counter() - t1) / sampled * 1e3
    # the same step with the first-level FPS inline (no next batch announced): what a loop that cannot look ahead gets -
    # and the like-for-like figure against round 1's line; outside the timed region, rank-local
    no_prefetch_ms = None
    if trainer.prefetch is not None and not use_dist:
        k2 = max(2, min(5, args.steps))
        trainer.train_step(batch)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(k2):
            trainer.train_step(batch)
        torch.cuda.synchronize()
        no_prefetch_ms = round((time.perf_counter() - t1) / k2 * 1e3, 3)
    # the same step on CHANGING data: two resident batches alternate, each announced one step ahead and staged through
    # the trainer's static buffers (clouds copied, label tensors by reference) - what a training loop over a dataset
    # gets; the headline loops on one resident batch (its sampling still runs every step)
    changing_ms = None
    if trainer.graph and not use_dist and not stress and os.environ.get("GB_BENCH_CHANGING", "1") != "0":
        other = make_training_batch([1000 * rank + 500 + i for i in range(BATCH_PER_GPU)], NUM_POINT, device=device)
        pair = [batch, other]
        k2 = max(4, min(8, args.steps))
        for i in range(3):
            trainer.train_step(pair[i % 2], next_batch=pair[(i + 1) % 2])
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(3, 3 + k2):
            trainer.train_step(pair[i % 2], next_batch=pair[(i + 1) % 2])
        torch.cuda.synchronize()
        changing_ms = round((time.perf_counter() - t1) / k2 * 1e3, 3)
        del other, pair
    peak_mem_gb = round(torch.cuda.max_memory_allocated(device) / 2 ** 30, 2)
    t = torch.tensor([elapsed, 1.0, host_enqueue], dtype=torch.float64, device=device)
    ranks_seen = 1
    allreduce = None
    if use_dist:
        dist.all_reduce(t[:1], op=dist.ReduceOp.MAX)
        dist.all_reduce(t[1:2], op=dist.ReduceOp.SUM)
        dist.all_reduce(t[2:], op=dist.ReduceOp.MAX)     # the slowest rank's host work: a host-bound rank shows here
        ranks_seen = int(t[1].item())
        # exposed: what the compute stream waited for inside the timed steps; standalone: the same collectives with
        # nothing to hide under (after the timed region)
        allreduce = {"exposed_ms": trainer.grads.exposed_ms(), "standalone_ms": trainer.grads.standalone_ms(),
                     "buckets": len(trainer.grads.flat), "bytes": 4 * int(trainer.grads.flat_all.numel()),
                     # the step's collectives in issue order as (first element, elements) of the flat fp32 gradient -
                     # the same on every rank and in every execution mode - and what the last replayed step enqueued
                     "schedule": [[int(a), int(n)] for a, n in trainer.grads.schedule()],
                     "enqueue_order": list(getattr(trainer, "enqueue_log", []))}
        # the replicas must hold the SAME parameters after the timed steps (identical averaged gradients, identical
        # updates): an fp64 checksum and a bit-pattern checksum per rank, compared through MIN / MAX all-reduces
        flat = torch.cat([p.detach().reshape(-1) for p in trainer.net.parameters()])
        chk = torch.stack([flat.double().sum(), flat.view(torch.int32).double().sum()])
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        allreduce["replicas_in_sync"] = bool(torch.equal(lo, hi))
    elapsed = float(t[0].item())
    host_enqueue_max = float(t[2].item())

    if rank == 0:
        clouds = world * BATCH_PER_GPU * args.steps
        # dominant kernels of the step by total time: the two fp32 MFMA GEMM kernels of the fused SharedMLP
        # path (gemm_cl_kernel: LDS-tiled; gemm_rs_kernel: row-streaming).  "roofline" is whichever has the
        # larger share, the other goes to "roofline_gemm2".  Per launch: algorithmic FLOP = 2*P*K*N;
        # achieved = sum FLOP / sum launch durations (HIP events on the launch stream, in the timed region).
        ev_bias = _lib.event_pair_overhead_ms(device)  # what an empty event bracket reads; removed from every launch

        def gemm_roofline(kernel, what):
            ev = [(max(a.elapsed_time(b) - ev_bias, 1e-4), m["flop"]) for n in gemm_names
                  for a, b, m in kt.events[n] if m["kernel"] == kernel]
            if not ev:
                return None
            ms = sum(t for t, _ in ev)
            flop = sum(f for _, f in ev)
            achieved = flop / (ms * 1e-3) / 1e12
            if prec == "bf16":
                # bf16 matrix cores (2.5 PFLOP/s dense) with fp32 tensors in memory: the contraction is bound by
                # reading X once and writing Y once - algorithmic bytes 4 (P K + P N + K N) per launch
                byt = sum(4.0 * (P * (K + N) + K * N) for n in gemm_names for a, b, m in kt.events[n] if m["kernel"] == kernel
                          for P, K, N in m.get("pkn_list", [m["pkn"]]))   # (a grouped wgrad launch lists its products)
                gbs = byt / (ms * 1e-3) / 1e9
                return {"kernel": "%s (v_mfma_f32_32x32x16_bf16, fp32 operands in HBM; %s)" % (kernel, what), "bound": "hbm",
                        "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                        "traffic": None, "launch_ms": round(ms / len(ev), 4), "launches": len(ev),
                        "ms_per_step": round(ms / sampled, 3), "tflops": round(achieved, 1),
                        "mfma_bf16_frac": round(achieved / MFMA_BF16_PEAK_TFLOPS, 4)}
            traffic, traffic_source = pmc_traffic(kernel, with_source=True)
            split = kernel in ("gemm_rs_kernel", "wgrad_direct_kernel") and prec == "f32" and _fm._SPLIT3
            isa = ("v_mfma_f32_32x32x2_f32, and v_mfma_f32_32x32x16_bf16 for the products run as three-way exact bf16 splits "
                   "(GB_PREC_F32_SPLIT3: fp32 MFMA's error against fp64; priced against the roof that binds - executed bf16 FLOP vs the "
                   "bf16 matrix cores, bytes vs HBM - with the fp32-equivalent figure beside it)"
                   if split else "v_mfma_f32_32x32x2_f32")
            out = {"kernel": "%s (%s; %s)" % (kernel, isa, what), "bound": "mfma",
                   "achieved": round(achieved, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                   "frac": round(achieved / MFMA_F32_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_source,
                   "launch_ms": round(ms / len(ev), 4), "launches": len(ev),
                   "ms_per_step": round(ms / sampled, 3), "gflop_per_launch": round(flop / len(ev) / 1e9, 3)}
            if split:
                # these kernels issue v_mfma_f32_32x32x16_bf16 six times per fp32 product, so the
                # fp32 MFMA peak no longer binds them (its "ceiling" would be 2500 / 6 / 157.3 = 2.65).  The roofs that do:
                # the bf16 matrix cores against the EXECUTED flop (6 x algorithmic; an upper bound on the executed share -
                # the few shapes no split instantiation fits run as fp32 MFMA) and HBM against the bytes per launch (the
                # PMC traffic while the committed counters match this source, the algorithmic 4 (P (K + N) + K N) bytes
                # otherwise).  `frac` is the larger of the two; the old figure stays as fp32_equivalent_frac.
                alg_bytes = sum(4.0 * (P * (K + N) + K * N) for n in gemm_names for a, b, m in kt.events[n]
                                if m["kernel"] == kernel for P, K, N in m.get("pkn_list", [m["pkn"]])) / len(ev)
                per_launch = traffic if traffic is not None else alg_bytes
                hbm_gbs = per_launch / (ms / len(ev) * 1e-3) / 1e9
                mfma_exec = 6.0 * achieved / MFMA_BF16_PEAK_TFLOPS
                hbm_frac = hbm_gbs / HBM_PEAK_GBS
                out["fp32_equivalent_frac"] = out["frac"]
                out["mfma_executed_frac"] = round(mfma_exec, 4)
                out["hbm_frac"] = round(hbm_frac, 4)
                out["hbm_frac_bytes"] = "pmc traffic" if traffic is not None else "algorithmic bytes (no valid counters for this source)"
                out["algorithmic_bytes_per_launch"] = round(alg_bytes, 1)
                if hbm_frac >= mfma_exec:
                    out.update({"bound": "hbm", "achieved": round(hbm_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": round(hbm_frac, 4)})
                else:
                    out.update({"bound": "mfma", "achieved": round(6.0 * achieved, 1), "peak": MFMA_BF16_PEAK_TFLOPS,
                                "unit": "TFLOP/s", "achieved_is": "EXECUTED bf16 FLOP (6 x the algorithmic fp32 FLOP)",
                                "frac": round(mfma_exec, 4)})
                out["tflops_fp32_equivalent"] = round(achieved, 2)
            return out

        if stress:
            out_metric = "point-clouds/sec fwd+bwd, 50k-pt stress scene (BASELINE configs[4])"
        else:
            out_metric = "point-clouds/sec fwd+bwd, 20k-pt GraspNet scene"
        rl_cl = gemm_roofline("gemm_cl_kernel", "register-staged LDS tiles: tall split-K wgrad, unaligned shapes")
        rl_rs = gemm_roofline("gemm_rs_kernel", "row-streaming: tall fwd+BN-stats and dgrad+BN-backward sums")
        rl_ring = gemm_roofline("gemm_ring_kernel", "LDS-DMA ring: the few-row fwd / dgrad / wgrad products")
        rl_wg = gemm_roofline("wgrad_direct_kernel", "register-direct: the tall wgrads, operands straight from HBM into the matrix cores")
        rl_grp = gemm_roofline("gemm_ring_group_kernel", "the step's few-row weight gradients recorded during backward and "
                               "run as grouped launches of up to 63 products on the ring kernel's 64 x 64 tiles")
        both = sorted([r for r in (rl_cl, rl_rs, rl_ring, rl_wg, rl_grp) if r], key=lambda r: -r["ms_per_step"])
        roofline = both[0] if both else None
        roofline_second = both[1] if len(both) > 1 else None
        roofline_third = both[2] if len(both) > 2 else None
        roofline_fourth = both[3] if len(both) > 3 else None
        roofline_fifth = both[4] if len(both) > 4 else None
        # largest single launch of the step: the first-level FPS (HBM class, streaming-model bytes)
        roofline_fps = None
        fps_ms = 0.0
        big = [(a.elapsed_time(b) - ev_bias, m) for a, b, m in kt.events["gb_fps"] if m["n"] == NUM_POINT]
        if big:
            fps_ms = sum(x for x, _ in big) / len(big)
            meta = big[0][1]
            fps_bytes = fps_algorithmic_bytes(meta["b"], meta["n"], meta["m"])
            achieved = fps_bytes / (fps_ms * 1e-3) / 1e9
            pruned = _lib._fps_prune and _lib.FPS_PRUNE_MIN_N <= meta["n"] <= _lib.FPS_PRUNE_MAX_N
            rows_kernel = meta["n"] <= 20480 and _lib._fps_layout != _lib.FPS_LAYOUT["r4"]
            kname = ("fps_rows_kernel" if rows_kernel else "fps_pruned_kernel") if pruned else "fps_reg_kernel<1024, 20>"
            order = {"rows": "row-order (two-level equal-count counting sort)", "cell": "cell-order counting sort",
                     "morton": "Morton keys + sort"}[_lib._fps_order]
            pk = ("fps_rows_kernel (run-time indexed row registers)" if rows_kernel else "fps_pruned_kernel<1024,20>") \
                if meta["n"] <= 20480 else "fps_pruned_big_kernel (rows streamed from a sorted copy)"
            what = order + " + " + pk if pruned else "fps_reg_kernel<1024,20>"
            roofline_fps = {"kernel": "%s (furthest_point_sampling %d->%d, b=%d)" % (what, meta["n"], meta["m"], meta["b"]),
                            "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(kname),
                            "launch_ms": round(fps_ms, 4), "launches": len(big)}
        # first-level ball query and the fused 16-fold cylinder query: neighbour scans over an L2-resident cloud, so the
        # streaming-byte model says nothing (it exceeds HBM peak); they are priced against the vector ALU instead -
        # lane-instructions per scanned pair x the scanned pairs the kernels report, over the launch time
        roofline_ball = roofline_cyl = roofline_fps_ball = None
        ball = [(a.elapsed_time(b) - ev_bias, m) for a, b, m in kt.events["gb_ball_query"] if m["n"] == NUM_POINT]
        if ball:
            ball_ms = sum(x for x, _ in ball) / len(ball)
            m = next(mm for _, mm in ball if "args" in mm)   # the first timed launch of the shape keeps its inputs
            new_xyz, xyz = m["args"]
            idx = torch.empty((m["b"], m["m"], m["ns"]), dtype=torch.int32, device=device)
            scanned = torch.empty((m["b"], m["m"]), dtype=torch.int32, device=device)
            _lib.check(_lib.lib().gb_ball_query(_lib.ptr(new_xyz), _lib.ptr(xyz), _lib.ptr(idx), _lib.ptr(scanned), m["b"],
                                                m["n"], m["m"], m["radius"], m["ns"], _lib.current_stream(device)), "ball")
            pairs = float(scanned.double().sum())
            ach = pairs * BALL_OPS_PER_PAIR / (ball_ms * 1e-3) / 1e12
            ball_bytes = 12.0 * pairs + 12.0 * m["b"] * m["m"] + 4.0 * m["b"] * m["m"] * m["ns"]
            roofline_ball = {"kernel": "ball_query_kernel (first level: %d centres x %d points, ns=%d, b=%d)"
                                       % (m["m"], m["n"], m["ns"], m["b"]), "bound": "valu",
                             "achieved": round(ach, 2), "peak": round(VALU_PEAK_TLANEOPS, 1), "unit": "T lane-ops/s",
                             "frac": round(ach / VALU_PEAK_TLANEOPS, 4), "scanned_pairs": pairs,
                             "scanned_frac_of_full": round(pairs / (m["b"] * m["m"] * float(m["n"])), 4),
                             "gpairs_per_s": round(pairs / (ball_ms * 1e-3) / 1e9, 1), "launch_ms": round(ball_ms, 4),
                             "launches": len(ball), "streaming_model_gbs": round(ball_bytes / (ball_ms * 1e-3) / 1e9, 1)}
            if roofline_fps:
                # north_star's "FPS + ball-query" figure: both first-level launches together, streaming-model bytes
                tot_ms = fps_ms + ball_ms
                ach = (fps_bytes + ball_bytes) / (tot_ms * 1e-3) / 1e9
                roofline_fps_ball = {"kernel": "first-level FPS + ball query together", "bound": "hbm",
                                     "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": round(ach / HBM_PEAK_GBS, 4), "launch_ms": round(tot_ms, 4)}
        cyl = [(a.elapsed_time(b) - ev_bias, m) for a, b, m in kt.events["gb_cylinder_query_multi"]]
        if cyl:
            cyl_ms = sum(x for x, _ in cyl) / len(cyl)
            m = next(mm for _, mm in cyl if "args" in mm)
            new_xyz, xyz, rot9 = m["args"]
            # the fused kernel scans a centre until the slowest of its 16 lists is full: max of the 16 single counts
            worst = torch.zeros((m["b"], m["m"]), dtype=torch.int32, device=device)
            idx = torch.empty((m["b"], m["m"], m["ns"]), dtype=torch.int32, device=device)
            scanned = torch.empty((m["b"], m["m"]), dtype=torch.int32, device=device)
            for r in m["radii"]:
                for h in m["hmaxs"]:
                    _lib.check(_lib.lib().gb_cylinder_query(_lib.ptr(new_xyz), _lib.ptr(xyz), _lib.ptr(rot9), _lib.ptr(idx),
                                                            _lib.ptr(scanned), m["b"], m["n"], m["m"], r, m["hmin"], h,
                                                            m["ns"], _lib.current_stream(device)), "cyl")
                    worst = torch.maximum(worst, scanned)
            pairs = float(worst.double().sum())
            ach = pairs * CYL16_OPS_PER_PAIR / (cyl_ms * 1e-3) / 1e12
            roofline_cyl = {"kernel": "cylinder_query_kernel<4,4> (16 queries fused: %d seeds x %d points, ns=%d, b=%d)"
                                      % (m["m"], m["n"], m["ns"], m["b"]), "bound": "valu", "achieved": round(ach, 2),
                            "peak": round(VALU_PEAK_TLANEOPS, 1), "unit": "T lane-ops/s",
                            "frac": round(ach / VALU_PEAK_TLANEOPS, 4), "scanned_pairs": pairs,
                            "scanned_frac_of_full": round(pairs / (m["b"] * m["m"] * float(m["n"])), 4),
                            "gpairs_per_s": round(pairs / (cyl_ms * 1e-3) / 1e9, 1), "launch_ms": round(cyl_ms, 4),
                            "launches": len(cyl)}
        out = {
            "metric": out_metric,
            "value": round(clouds / elapsed, 3), "unit": "point-clouds/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": prec,
            "data": "synthetic (make_scene clouds + seeded uniform grasp labels; no dataset available)" +
                    (" - REHEARSAL: all ranks on one GPU over gloo, not a measurement" if rehearsal else ""),
            "config": {"workload": "%s: GraspBalance train step fwd+bwd+Adam, B=%d/GPU, N=%d points, "
                                   "8 objects x 300 grasp points x 300 views labels%s"
                                   % ("configs[4] (stress)" if stress else "configs[3]", BATCH_PER_GPU, NUM_POINT,
                                      ", bf16 MLP contractions / fp32 geometry, statistics and storage" if stress else ""),
                       "global_batch": world * BATCH_PER_GPU, "parallelism": "dp%d" % world},
            "roofline": roofline,
            "roofline_gemm2": roofline_second,
            "roofline_gemm3": roofline_third,
            "roofline_gemm4": roofline_fourth,
            "roofline_gemm5": roofline_fifth,
            "fp32_products": ("tall products (row-streaming forward / dgrad, register-direct wgrad) as three-way exact bf16 splits on the matrix cores (GB_PREC_F32_SPLIT3: "
                              "six bf16 products per fp32 product, fp32 accumulation, same error against fp64 as fp32 MFMA - "
                              "tests/test_gemm_gpu.py, tools/split3_probe.hip); every other product fp32 MFMA; GB_SPLIT3=0 "
                              "runs all of them on fp32 MFMA") if (prec == "f32" and _fm._SPLIT3) else "fp32 MFMA",
            "event_bias_us": round(ev_bias * 1e3, 2),
            "event_sampled_steps": sampled,
            "roofline_fps": roofline_fps,
            "roofline_ball": roofline_ball,
            "roofline_fps_ball": roofline_fps_ball,
            "roofline_cyl": roofline_cyl,
            "ranks_seen": ranks_seen,
            "prefetch_sampling": trainer.prefetch is not None,
            "ms_per_step_no_prefetch": no_prefetch_ms, "host_enqueue_ms_per_step": round(host_enqueue / args.steps * 1e3, 3),
            "host_enqueue_ms_per_step_max_over_ranks": round(host_enqueue_max / args.steps * 1e3, 3),
            "host_waiting_for_gpu_ms_per_step": round(host_waited / args.steps * 1e3, 3),
            "execution": ("hip-graph replay: the step captured once (warm-up), %d replays timed, %d graph(s) captured"
                          % (args.steps, len(trainer._graphs))) if trainer.graph
                         else ("eager: every launch of every step enqueued by the host"
                               + (" (fallback: %s)" % graph_fallback if graph_fallback else "")),
            # launch-by-launch execution as shipped (Trainer(graph=False), nothing instrumented): the guard's probe above
            "ms_per_step_eager": round(t_eager_clean, 3) if t_eager_clean is not None else None,
            # (the leg the per-kernel rooflines are measured on: HIP events around every launch of interest - NOT a
            # statement about launch-by-launch execution)
            "ms_per_step_instrumented_leg": round(eager_ms, 3),
            "ms_per_step_changing_data": changing_ms,
            "max_memory_allocated_gb": peak_mem_gb,
            "cpu_affinity": affinity,
        }
        for r in (out["roofline"], out["roofline_gemm2"], out["roofline_gemm3"], out["roofline_gemm4"], out["roofline_gemm5"], out["roofline_fps"], out["roofline_ball"], out["roofline_fps_ball"],
                  out["roofline_cyl"]):
            if r:
                r["measured_on"] = ("%d eager steps of the same trainer right after the timed region (HIP events around every "
                                    "launch; the timed steps replay a graph)" % sampled)
        if allreduce:
            out["allreduce_ms"] = round(allreduce["standalone_ms"], 4)
            out["allreduce_exposed_ms"] = round(allreduce["exposed_ms"], 4)
            out["replicas_in_sync"] = allreduce["replicas_in_sync"]
            out["allreduce"] = {"buckets": allreduce["buckets"], "bytes": allreduce["bytes"],
                                "schedule": allreduce["schedule"], "enqueue_order": allreduce["enqueue_order"],
                                "note": "allreduce_ms = the step's bucket all-reduces back to back with nothing to hide "
                                        "under (median of 5 after the timed region); allreduce_exposed_ms = mean time per "
                                        "timed step the compute stream waited for them (HIP events around the waits)"}
        if world == 1 and not stress and not args.no_extra_configs:
            # BASELINE configs[2] and configs[4] in the same line (short legs, a few seconds each): their own ms_per_step and
            # dominant-kernel roofline.  The train trainer's memory goes first (configs[4] needs 19 GB of its own).
            del trainer, batch, kt, timer, loss
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            out["configs"] = {"infer": _config_leg(["--config", "infer", "--steps", "6", "--warmup", "2"]),
                              "stress": _config_leg(["--config", "stress", "--steps", "3", "--warmup", "1",
                                                     "--no-cpu-baseline", "--no-extra-configs"])}
        if world == 1 and not args.no_cpu_baseline and not stress:
            threads = min(os.cpu_count() or 1, 32)
            out["cpu_baseline"] = cpu_baseline(threads, device)
            out["cpu_baseline_dense"] = cpu_baseline_dense(threads, device)
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
