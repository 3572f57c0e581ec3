"""world_size-2 gloo rehearsal of the RCCL data-parallel path (graspbalance_amd/data_parallel.py):
flat-bucket gradient all-reduce == mean of the per-rank gradients, replicas stay identical after the
optimizer step, BatchNorm statistics stay per rank, nested ``*_list`` labels are sharded by item."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests import cpu_backend
    from tests.test_model_cpu import _tiny_batch, _tiny_net
    from graspbalance_amd.data_parallel import FlatGradAllReduce, broadcast_module, shard_batch
    from graspbalance_amd.loss import get_loss
    cpu_backend.install()
    net = _tiny_net()
    if rank == 1:  # de-synchronise on purpose: broadcast must repair it
        with torch.no_grad():
            for p in net.parameters():
                p.add_(1.0)
    broadcast_module(net)
    net.train()
    full = _tiny_batch(B=2)
    mine = shard_batch(full, rank, world)
    assert mine['point_clouds'].shape[0] == 1 and len(mine['grasp_points_list']) == 1
    # the pairing Trainer(distributed=True) builds: FlatAdam's gradient buffer IS the all-reduce buffer
    from graspbalance_amd.flat_adam import FlatAdam
    opt = FlatAdam(net.parameters(), lr=1e-3)
    grads = FlatGradAllReduce(net, bucket_mb=0.5, flat=(opt._flat_g, opt._grad_views, opt._params))
    assert len(grads.buckets) > 1
    assert sum(f.numel() for f in grads.flat) == opt._flat_g.numel()
    loss, _ = get_loss(net(mine))
    loss.backward()
    local = [p.grad.clone() if p.grad is not None else torch.zeros_like(p) for p in net.parameters()]
    assert grads._next >= 1, "no bucket was all-reduced during backward (overlap hooks did not fire)"
    grads.reduce()
    reduced = [p.grad.clone() for p in net.parameters()]
    lo, hi = opt._flat_g.data_ptr(), opt._flat_g.data_ptr() + 4 * opt._flat_g.numel()
    assert all(lo <= p.grad.data_ptr() < hi for p in net.parameters())  # .grad = views of the optimizer's buffer
    opt.step()
    torch.save({"local": local, "reduced": reduced, "params": [p.detach().clone() for p in net.parameters()],
                "bn_mean": net.view_estimator.FeatureExtraction.sa1.mlp_module.layer0.bn.bn.running_mean.clone()},
               os.path.join(out_dir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_flat_bucket_allreduce_world2(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(os.path.join(tmp_path, "rank0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "rank1.pt"))
    for a, b, m0, m1 in zip(r0["local"], r1["local"], r0["reduced"], r1["reduced"]):
        assert torch.equal(m0, m1)  # every rank holds the same reduced gradient
        assert torch.allclose(m0, (a + b) / 2, rtol=1e-6, atol=1e-8)
    assert any(not torch.equal(a, b) for a, b in zip(r0["local"], r1["local"]))  # shards really differ
    for p0, p1 in zip(r0["params"], r1["params"]):
        assert torch.equal(p0, p1)  # replicas identical after the step
    assert not torch.equal(r0["bn_mean"], r1["bn_mean"])  # BatchNorm statistics are per rank (no SyncBN)


def _schedule_worker(rank, world, port, out_dir, modes):
    """One train step per entry of `modes` from identical state: 'hooks' = launch-by-launch execution (post-accumulate
    hooks issue a bucket when it is complete, reduce() the rest), 'split' = what the HIP-graph step does (gradient cut,
    backward in two parts, each part packs its slice, issue_packed / reduce_flat between them), 'whole' = no cut, one
    bucket.  modes[k][rank] is what THIS rank does in round k - ranks may differ within a round."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests import cpu_backend
    from tests.test_model_cpu import _tiny_batch, _tiny_net
    from graspbalance_amd.data_parallel import FlatGradAllReduce, shard_batch
    from graspbalance_amd.drp import GRAD_CUT, GRAD_CUT_IN, GRAD_CUT_OUT, grad_cut_param_index
    from graspbalance_amd.flat_adam import FlatAdam
    from graspbalance_amd.loss import get_loss
    cpu_backend.install()
    mine = shard_batch(_tiny_batch(B=max(2, world)), rank, world)
    results = []
    for round_modes in modes:
        mode = round_modes[rank]
        net = _tiny_net()
        net.train()
        opt = FlatAdam(net.parameters(), lr=1e-3)
        cut = None if all(m == "whole" for m in round_modes) else grad_cut_param_index(net)
        assert all(m == "whole" for m in round_modes) or cut is not None
        grads = FlatGradAllReduce(net, bucket_mb=1e9, flat=(opt._flat_g, opt._grad_views, opt._params), cut=cut)
        sched = grads.schedule()
        total = opt._flat_g.numel()
        assert sum(n for _, n in sched) == total and (len(sched) == 2 if cut is not None else len(sched) == 1)
        if cut is not None:  # deep slice first: the tail of the registration order, the bulk of the parameters
            assert sched[0][0] + sched[0][1] == total and sched[1][0] == 0 and sched[0][1] > 0.8 * total
        if mode == "hooks":
            loss, _ = get_loss(net(dict(mine)))
            loss.backward()
            if cut is not None:
                assert grads._next >= 1, "the deep slice was not issued during backward"
            grads.reduce()
        else:
            inputs = dict(mine)
            if mode == "split":
                inputs[GRAD_CUT] = True
            end_points = net(inputs)
            cut_t = (end_points[GRAD_CUT_OUT], end_points[GRAD_CUT_IN]) if mode == "split" else None
            loss, _ = get_loss(end_points)
            grads.hold = True
            loss.backward()
            if mode == "split":
                n_deep = sum(int(p.grad is not None) for p in opt._params[cut:])
                assert n_deep == len(opt._params) - cut and all(p.grad is None for p in opt._params[:cut])
                opt.pack(cut, None)
                grads.issue_packed(0)
                cut_t[0].backward(cut_t[1].grad)
                opt.pack(0, cut)
            else:
                opt.pack()
            grads.hold = False
            grads.reduce_flat()
        reduced = opt._flat_g.clone()
        opt.step(packed=(mode != "hooks"))
        results.append({"mode": mode, "reduced": reduced, "params": opt._flat_p.clone(), "schedule": sched})
        dist.barrier()
    torch.save(results, os.path.join(out_dir, "rank%d.pt" % rank))
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_one_collective_schedule_for_every_execution_mode_world2(tmp_path):
    """ADVICE round 4 / VERDICT round 4 #2: the ranks of one step may execute it differently (one replays graphs - the
    split backward with the first slice's all-reduce beside the second part - while another runs launch by launch
    because its batch signature is new or over the limit) and still issue the same collectives; the split path gives
    bit-identical gradients and parameters to the launch-by-launch path and to ONE collective over the whole buffer."""
    port = _free_port()
    modes = [("hooks", "hooks"), ("split", "split"), ("hooks", "split"), ("split", "hooks"), ("whole", "whole")]
    mp.spawn(_schedule_worker, args=(2, port, str(tmp_path), modes), nprocs=2, join=True)
    r0 = torch.load(os.path.join(tmp_path, "rank0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "rank1.pt"))
    for a, b in zip(r0, r1):
        assert a["schedule"] == b["schedule"]
        assert torch.equal(a["reduced"], b["reduced"]) and torch.equal(a["params"], b["params"])
    for k in range(1, len(modes)):   # every round started from the same state: same reduced gradient, same parameters
        assert torch.equal(r0[k]["reduced"], r0[0]["reduced"]), modes[k]
        assert torch.equal(r0[k]["params"], r0[0]["params"]), modes[k]
    assert float(r0[0]["reduced"].abs().sum()) > 0


@pytest.mark.timeout(1200)
def test_one_collective_schedule_world4_with_one_rank_running_launch_by_launch(tmp_path):
    """four ranks, and in one step ONE of them executes launch by launch (post-accumulate hooks) while
    the other three run the split backward of the HIP-graph step - what happens when a rank's batch signature is new or
    beyond GB_GRAPH_MAX_SIGNATURES while its peers replay.  Same collectives in the same order on every rank, reduced
    gradients and parameters bit-identical across ranks and equal to the all-hooks and all-split executions."""
    port = _free_port()
    modes = [("hooks",) * 4, ("split", "split", "hooks", "split"), ("split",) * 4]
    mp.spawn(_schedule_worker, args=(4, port, str(tmp_path), modes), nprocs=4, join=True)
    rs = [torch.load(os.path.join(tmp_path, "rank%d.pt" % r)) for r in range(4)]
    for k in range(len(modes)):
        for r in range(1, 4):
            assert rs[r][k]["schedule"] == rs[0][k]["schedule"] and len(rs[0][k]["schedule"]) == 2
            assert torch.equal(rs[r][k]["reduced"], rs[0][k]["reduced"]) and torch.equal(rs[r][k]["params"], rs[0][k]["params"])
        assert torch.equal(rs[0][k]["reduced"], rs[0][0]["reduced"]) and torch.equal(rs[0][k]["params"], rs[0][0]["params"]), modes[k]
    assert [x["mode"] for x in rs[2]] == ["hooks", "hooks", "split"] and float(rs[0][0]["reduced"].abs().sum()) > 0


def test_shard_batch_chunks_like_list_scatter():
    from graspbalance_amd.data_parallel import shard_batch
    batch = {"point_clouds": torch.arange(5).view(5, 1), "grasp_points_list": [[i] for i in range(5)]}
    sizes = [len(shard_batch(batch, r, 2)["grasp_points_list"]) for r in range(2)]
    assert sizes == [3, 2]  # chunk = ceil(5/2), the reference's list_scatter rule (data_parallel.py:31)
    assert shard_batch(batch, 1, 2)["point_clouds"].flatten().tolist() == [3, 4]
