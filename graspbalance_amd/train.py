"""Training harness with the loop semantics of the reference's train.py (Adam :94, OneCycleLR
:107-108, BN-momentum schedule :110-113, per-batch body :138-155) on the GraspBalance model, plus the
RCCL data-parallel gradient reduction.  (The reference script itself cannot run as shipped: it imports
the absent ``graspnet.GraspNet_MSCQ`` and dataset class names that do not exist — SURVEY.md §0.)

HIP-graph execution (``Trainer(graph=True)``, the default on a GPU).  A train step is ~1000 kernel launches and 15-19 ms
of Python to enqueue them - as long as the GPU itself needs.  Nothing in the step depends on a value the host has to
see (the one row count that used to be read back stays on the device: fused_mlp.set_static_rows), so the whole step -
forward, label matching, loss, backward, Adam - is captured ONCE into a HIP graph and replayed: one launch call per step
(plus, on its side stream, the three launches of the next batch's first-level sampling).  What varies between steps lives in device memory the
graph reads: the batch (static input buffers: ``Trainer.resident``), the learning rate (a device scalar), Adam's step
count.  A change of anything that was baked into launch arguments (tensor shapes of the batch, the BatchNorm momentum
of the epoch, whether a next batch is announced) captures another graph; all graphs share one memory pool.  With more
than one rank the gradient all-reduce sits BETWEEN two graphs (forward + backward + pack | RCCL all-reduce of the flat
buckets | Adam): collectives are not captured.
"""
import contextlib

import time
import weakref

import torch
from torch.optim.lr_scheduler import OneCycleLR

from . import fused_mlp
from .data_parallel import FlatGradAllReduce, broadcast_module
from .flat_adam import FlatAdam
from .graspbalance import GraspBalance
from .label_generation import LEAN
from .loss import get_loss
from .pytorch_utils import BNMomentumScheduler

import os
_PREFETCH_AT = os.environ.get("GB_PREFETCH_AT", "sa1")  # A/B switch: "start" | "sa1" | "off"
_GRAPH_DEFAULT = os.environ.get("GB_GRAPH", "1") != "0"  # A/B switch: 0 = every step enqueued launch by launch
# Where the graph step launches the next batch's first-level sampling (2.1 ms of one workgroup per cloud on a side stream):
# "bwd" = between the forward graph and the backward graph, i.e. beside the start of the backward; "start" = before the
# forward graph.  Measured (tools/fps_interference.py, B = 4 x 20 000): the step's own kernels take 17.2 ms; with the
# sampling beside the START of the forward 18.5 ms - the first levels' few-row GEMMs are one-tile-per-CU grids, and every
# one of them waits for the tile that landed on a CU the sampling occupies; beside the backward's first kernels (split-K
# grids of ~1000 workgroups, persistent row-streaming grids sized for the CUs left over) see DESIGN.md section 5.6.
_SAMPLE_AT = os.environ.get("GB_SAMPLE_AT", "bwd")
_MAX_SIGNATURES = int(os.environ.get("GB_GRAPH_MAX_SIGNATURES", "4"))
_HOST_SIDE_ORDER = os.environ.get("GB_HOST_SIDE_ORDER", "1") != "0"   # A/B switch: 0 = the side stream waits on the GPU
_LABEL_TABLES = os.environ.get("GB_LABEL_TABLES", "1") != "0"   # A/B switch: 0 = the label tensors are copied into static buffers
_LABEL_CAPACITY = os.environ.get("GB_LABEL_CAPACITY", "1") != "0"   # A/B switch: 0 = a captured step is keyed on every label tensor's shape
MAX_LABEL_SOURCES = 128   # entries of the label kernels' source-pointer tables (csrc/group.hip LG_MAX_SRC)
_NO_CONTEXT = contextlib.nullcontext()

BN_MOMENTUM_INIT = 0.5
BN_MOMENTUM_MAX = 0.001


class Trainer:
    def __init__(self, device, learning_rate=0.001, weight_decay=0.0, bn_decay_step=2, bn_decay_rate=0.5,
                 steps_per_epoch=100, max_epoch=18, num_view=300, seed=1234, distributed=False,
                 bucket_mb=16.0, model=None, time_collectives=False, prefetch_sampling=True, mlp_precision=None,
                 lean_labels=True, graph=None):
        torch.manual_seed(seed)
        self.device = torch.device(device)
        # 'bf16': BASELINE configs[4].  A property of THIS trainer: every step runs inside fused_mlp.precision(...), the
        # GEMM calls carry it (GbGemmOpts), so two trainers of one process may differ
        self.mlp_precision = mlp_precision if self.device.type == "cuda" else None
        self.lean_labels = bool(lean_labels) and os.environ.get("GB_LEAN_LABELS", "1") != "0"  # A/B switch
        self.net = model if model is not None else GraspBalance(
            input_feature_dim=0, num_view=num_view, num_angle=12, num_depth=4, cylinder_radius=0.08,
            hmin=-0.02, hmax_list=[0.01, 0.02, 0.03, 0.04])
        self.net.to(self.device)
        if distributed:
            broadcast_module(self.net)
        # the reference's optimizer (train.py:94: Adam, default betas / eps) on one flat parameter buffer: one fused
        # launch over 9 M elements instead of torch's multi-tensor lists over 253 tensors (flat_adam.py)
        self.optimizer = FlatAdam(self.net.parameters(), lr=learning_rate, weight_decay=weight_decay)
        self.scheduler = OneCycleLR(self.optimizer, max_lr=learning_rate, steps_per_epoch=steps_per_epoch,
                                    epochs=max_epoch)
        self._max_lr, self._max_epoch = learning_rate, max_epoch
        bn_lbmd = lambda it: max(BN_MOMENTUM_INIT * bn_decay_rate ** (int(it / bn_decay_step)), BN_MOMENTUM_MAX)
        self.bnm_scheduler = BNMomentumScheduler(self.net, bn_lambda=bn_lbmd, last_epoch=-1)
        # HIP-graph replay of the step (module docstring); needs the sync-free step (static rows) and a GPU
        self.graph = (_GRAPH_DEFAULT if graph is None else bool(graph)) and self.device.type == "cuda"
        # the stream the step is captured on.  Every parameter's AccumulateGrad node is created HERE, with that stream
        # current, and kept alive: such a node runs on the stream it was created under for as long as it lives, and one
        # born on the default stream (any earlier forward whose loss is still referenced keeps it alive) drags the
        # default stream into the capture - a fork the capture never joins (hipStreamEndCapture crashes on it)
        self._cstream = torch.cuda.Stream(device=self.device) if self.graph else None
        if self.graph:
            with torch.cuda.stream(self._cstream):
                self._acc_nodes = [p.expand_as(p).grad_fn.next_functions[0][0] for p in self.net.parameters()
                                   if p.requires_grad]
            torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)   # (eager steps on other streams)
        # the buckets ARE slices of the optimizer's flat gradient buffer: the averaged gradient lands where the
        # update reads it (no second 36 MB copy per step)
        # where the data-parallel step cuts its backward in two (drp.GRAD_CUT): the parameters behind the cut - 94 % of
        # them - have their gradients first, and their all-reduce runs under the rest of the backward
        from .drp import grad_cut_param_index
        self._cut = grad_cut_param_index(self.net) if distributed and os.environ.get("GB_GRAD_CUT", "1") != "0" else None
        with (torch.cuda.stream(self._cstream) if self._cstream is not None else _NO_CONTEXT):
            # no cut (GB_GRAD_CUT=0, a net without the DRP backbone): a graph step's collectives run behind its whole
            # backward, so it issues ONE over the flat buffer; only a launch-by-launch trainer has hooks to overlap
            # size-capped buckets with.  Decided by construction-time state that is the same on every rank.
            self.grads = FlatGradAllReduce(self.net, bucket_mb=(None if (self._cut is None and self.graph) else bucket_mb),
                                           timing=time_collectives,
                                           flat=(self.optimizer._flat_g, self.optimizer._grad_views,
                                                 self.optimizer._params), cut=self._cut)
        # the stream the first slice's all-reduce runs on beside the second part of the backward (graph execution)
        self._comm = torch.cuda.Stream(device=self.device) if (distributed and self.device.type == "cuda") else None
        self._cut_event = torch.cuda.Event() if self._comm is not None else None
        self.bnm_scheduler.step()
        self.net.train()
        # first-level FPS of the next batch on a side stream (prefetch.py); needs the caller to pass `next_batch`
        self.prefetch = None
        sa1 = getattr(getattr(getattr(self.net, "view_estimator", None), "FeatureExtraction", None), "sa1", None)
        if prefetch_sampling and _PREFETCH_AT != "off" and self.device.type == "cuda" and sa1 is not None and sa1.npoint:
            from .prefetch import SamplingPrefetch
            self.prefetch = SamplingPrefetch(self.device, sa1.npoint)
        self.distributed = bool(distributed)
        self._graphs = {}        # signature -> _StepGraph
        self._pool = None        # memory pool shared by the graphs (one replays at a time)
        self._static = None      # _StaticBatch: the input buffers the graphs of the current batch signature read
        self._statics = {}       # signature -> _StaticBatch (a graph reads the buffers it was captured on: kept with it)
        self._eager_signatures = set()
        self.graph_replays = 0
        self.enqueue_log = []    # the last replayed step's enqueue order (see _graph_step)

    def train_step(self, batch, next_batch=None):
        """forward -> loss -> backward -> gradient all-reduce -> Adam step -> LR step.  Returns the
        loss tensor (no host sync here; the reference's per-key .item() logging is the caller's).
        next_batch: the batch of the FOLLOWING call, if the loop already holds it - its first-level furthest-point
        sampling then runs on a side stream under this step (prefetch.py)."""
        with (fused_mlp.precision(self.mlp_precision) if self.mlp_precision is not None else _NO_CONTEXT):
            if self.graph:
                return self._graph_step(batch, next_batch)
            return self._train_step(batch, next_batch)

    def train_step_eager(self, batch, next_batch=None):
        """The same step enqueued launch by launch (what graph=False does): for tools that bracket single launches with
        events (bench.py's roofline leg), and the reference point of the graph's parity test."""
        with (fused_mlp.precision(self.mlp_precision) if self.mlp_precision is not None else _NO_CONTEXT):
            return self._train_step(batch, next_batch)

    def _train_step(self, batch, next_batch=None):
        if self.device.type == "cuda":
            fused_mlp.begin_step(self.device)  # one re-zeroed arena for the step's small fp64 reduction buffers
        inputs = dict(batch)  # the network adds its outputs to the dict it is given
        if self.lean_labels:
            inputs[LEAN] = True  # label matching builds only what this step reads (label_generation._lean_labels)
        if self.prefetch is not None:
            from .prefetch import AFTER_SA1, KEY
            inds = self.prefetch.take(batch['point_clouds'])
            if inds is not None:
                inputs[KEY] = inds
            if next_batch is not None:
                clouds = next_batch['point_clouds']
                if _PREFETCH_AT == "start":
                    self.prefetch.launch(clouds)
                else:
                    inputs[AFTER_SA1] = lambda: self.prefetch.launch(clouds)
        end_points = self.net(inputs)
        if self.prefetch is not None and next_batch is not None and self.prefetch.pending is None:
            self.prefetch.launch(next_batch['point_clouds'])  # a backbone without the hook: start it now
        loss, end_points = get_loss(end_points)
        with self._wgrad_queue(hooks_live=True):
            loss.backward()
        self.grads.reduce()
        self.optimizer.step()
        self.grads.zero_grad()  # .grad = None: the next backward assigns instead of accumulating
        self.scheduler.step()
        return loss

    # ---- checkpoints (the reference's dictionary: train.py:96-103, 226-234) --------------------------------------------
    def save_checkpoint(self, path, epoch, loss=None):
        """torch.save of {'epoch', 'optimizer_state_dict', 'loss', 'model_state_dict'} - the keys the reference writes
        after every epoch; the optimizer entry has torch.optim.Adam's layout (FlatAdam.state_dict), so the file loads
        into the reference's own script and vice versa."""
        torch.save({'epoch': int(epoch), 'optimizer_state_dict': self.optimizer.state_dict(),
                    'loss': float(loss.detach() if torch.is_tensor(loss) else loss) if loss is not None else None,
                    'model_state_dict': self.net.state_dict()}, path)

    def load_checkpoint(self, path):
        """Restore model and optimizer from a file of that layout and put the OneCycle schedule where the reference puts
        it on resume (last_epoch = start_epoch * steps_per_epoch - 1, train.py:107-108) -> start_epoch."""
        ckpt = torch.load(path, map_location=self.device)
        self.net.load_state_dict(ckpt['model_state_dict'])
        self.optimizer.load_state_dict(ckpt['optimizer_state_dict'])
        fused_mlp.invalidate_eval_tables()
        start_epoch = int(ckpt.get('epoch', 0))
        total = self.scheduler.total_steps
        per_epoch = max(1, total // max(1, getattr(self, "_max_epoch", 1)))
        self.scheduler = OneCycleLR(self.optimizer, max_lr=self._max_lr, total_steps=total,
                                    last_epoch=start_epoch * per_epoch - 1)
        self.bnm_scheduler.step(start_epoch)   # (the reference steps it once per epoch: train.py:136)
        return start_epoch

    # ---- HIP-graph execution ----------------------------------------------------------------------------------------
    def resident(self, batch):
        """Copy `batch` into this trainer's static input buffers and return the batch made of THOSE tensors: steps on it
        replay the graph without any staging copy (a data loader would write its next batch into them directly).  The
        large grasp label / offset / tolerance tensors are NOT copied when the lean label matching can read them through
        device-side pointer tables (_StaticBatch, by_reference): the returned batch holds the caller's own tensors for
        those keys, and they must stay unchanged while steps on them are in flight."""
        if not self.graph:
            return batch
        st = self._static_for(self._sig(batch), batch)
        if st is None:
            return batch
        st.load(batch)
        return st.batch

    def _capacity_form(self, batch):
        """Can this batch's label lists be held at capacity (label_generation.LabelGeometry)?  What the lean label
        matching needs of them anyway + the switch."""
        from .label_generation import LIST_KEYS, tables_ok, label_needs
        if not (self.lean_labels and _LABEL_TABLES and _LABEL_CAPACITY and all(k in batch for k in LIST_KEYS)
                and tables_ok(batch)):
            return False
        # the capacity form addresses the pointer tables at slot b*kc + j with kc >= the LARGEST cloud's object count, so
        # it needs B*max_per_cloud table entries where the packed form needs the total: an unbalanced batch (40+10+10+10
        # objects at B = 4) fits the tables' 128 entries packed and not at capacity -> it keeps the shape-keyed form
        # (csrc/group.hip LG_MAX_SRC)
        return len(batch['grasp_points_list']) * label_needs(batch)[0] <= MAX_LABEL_SOURCES

    def _sig(self, batch):
        """What a captured step is keyed on.  Capacity form: the label lists' own shapes are NOT part of it (only their
        common (V, A, D)), so batches whose objects / grasp points differ in number share buffers and graphs."""
        if self._capacity_form(batch):
            from .label_generation import LIST_KEYS
            return ("capacity", tuple(batch['grasp_labels_list'][0][0].shape[1:4])) + _signature(batch, skip=LIST_KEYS)
        return _signature(batch)

    def _static_for(self, sig, batch):
        """The static buffers (and with them the captured graphs) of a batch signature; None once more than
        GB_GRAPH_MAX_SIGNATURES (default 4) different signatures have been seen - every one costs a capture (~1.5 s), its
        own copy of the inputs and graph memory, so a loader whose shapes keep changing runs launch by launch instead."""
        capacity = None
        if sig and sig[0] == "capacity":
            # any existing set of buffers of this signature that is large enough serves the batch; a new one is sized
            # with room to spare (objects per cloud to a multiple of 4 - at most 128 per batch, the pointer tables'
            # limit -, grasp points per object to a power of two) so that the next larger scene fits it too
            from .label_generation import label_needs
            need_k, need_p = label_needs(batch)
            fit = [v for k, v in self._statics.items() if k[0] == sig and v.geometry.fits(batch)]
            if fit:
                self._static = min(fit, key=lambda v: v.geometry.kc * v.geometry.pc)
                return self._static
            B = len(batch['grasp_points_list'])
            kc = min(max(4, -(-need_k // 4) * 4), max(need_k, 128 // max(B, 1)))
            pc = 64
            while pc < need_p:
                pc *= 2
            capacity = (kc, pc)
            sig = (sig, kc, pc)
        st = self._statics.get(sig)
        if st is None:
            if sig in self._eager_signatures:
                return None
            if len(self._statics) >= _MAX_SIGNATURES:
                if not self._eager_signatures:
                    import warnings
                    warnings.warn("graspbalance_amd.Trainer: more than %d batch signatures (shapes) seen; further new ones "
                                  "run launch by launch instead of being captured" % _MAX_SIGNATURES)
                self._eager_signatures.add(sig)
                return None
            from .label_generation import tables_ok, BY_REFERENCE
            by_ref = BY_REFERENCE if (self.lean_labels and _LABEL_TABLES and tables_ok(batch)) else ()
            st = self._statics[sig] = _StaticBatch(batch, self.prefetch.npoint if self.prefetch is not None else 0, by_ref,
                                                   capacity=capacity)
        self._static = st
        return st

    def _wgrad_queue(self, hooks_live):
        """The context a backward of this trainer runs in: a fused_mlp.WgradQueue on the current stream (the few-row
        weight gradients recorded and launched together when it ends), or nothing - on the CPU, with GB_WGRAD_GROUP=0,
        and whenever post-accumulate hooks may issue a bucket's all-reduce from INSIDE the backward (launch-by-launch
        data-parallel execution: a hook would pack gradients the grouped launch has not written yet)."""
        import torch.distributed as dist
        if (self.device.type != "cuda" or not fused_mlp._WGRAD_GROUP
                or (hooks_live and self.distributed and dist.is_available() and dist.is_initialized())):
            return _NO_CONTEXT
        return fused_mlp.WgradQueue(self.device, params=self.grads.params)

    def _bn_momentum(self):
        for m in self.net.modules():
            if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d, torch.nn.BatchNorm3d)):
                return m.momentum
        return None

    def _graph_step(self, batch, next_batch):
        announced = self.prefetch is not None and next_batch is not None
        sig = self._sig(batch)
        st = self._static_for(sig, batch)
        if st is None:
            return self._train_step(batch, next_batch)
        st.load(batch)                                   # no copies when `batch` is st.batch (Trainer.resident)
        if announced:
            st.load_next(next_batch['point_clouds'])
        if announced and st.samp_for != st.cloud_id():
            # the graph takes the current batch's first-level samples from st.inds: nobody sampled this batch ahead
            with torch.no_grad():
                from . import pointnet2_utils
                st.inds.copy_(pointnet2_utils.furthest_point_sample(st.batch['point_clouds'][..., 0:3].contiguous(),
                                                                   self.prefetch.npoint))
            st.samp_for = st.cloud_id()
        key = (id(st), announced, self._bn_momentum(), self.mlp_precision, fused_mlp.get_precision())
        g = self._graphs.get(key)
        if g is None:
            g = self._graphs[key] = self._capture(st, announced)
        self.optimizer.set_lr_tensor()
        cur = torch.cuda.current_stream(self.device)
        stage_event = None
        if announced and _HOST_SIDE_ORDER:
            stage_event = st.stage_event      # everything the sampling reads has been enqueued on `cur` by now
            stage_event.record(cur)

        def sample_next():
            # The next batch's first-level sampling on the side stream.  Not inside a graph: a graph with a forked branch
            # loses ROCm's fast launch path for linear graphs (measured: hipGraphLaunch 14.6 ms of host time per replay
            # with the branch, 0.33 ms without), so it goes out beside two linear graphs.
            # And WITHOUT making the side stream wait on the main stream where that can be avoided: on this stack a stream
            # that waits on an event recorded in front of the step costs the step 0.75 ms (tools/fps_interference.py: 17.3
            # -> 18.05 ms with nothing at all running on the waiting stream).  The ordering the sampling needs is kept by
            # other means: its outputs rotate through RING slots, and a slot is reused only after the host has seen the
            # end of the step that read it three steps ago (an event that old has practically always fired: no stall).
            side = self.prefetch.side
            slot = st.ring_next()
            ver = st.batch['point_clouds']._version
            if not _is_self(st.next_src) or st.side_saw != ver:
                # the announced clouds were staged by the main stream this step (or the resident ones were written since
                # the side stream last synchronised with it): a real dependency.  Kept on the HOST where possible: a wait
                # packet that sits unsatisfied in the side queue while the main queue works is what costs the 0.7 ms
                # (held until the event has fired, the same wait is free - tools/fps_interference.py, "host at most 0
                # steps ahead").  The forward graph of this step is already enqueued, so the GPU does not idle while the
                # host waits here for the previous step to end.
                if _SAMPLE_AT != "start" and stage_event is not None:
                    if not stage_event.query():
                        t0 = time.perf_counter()
                        stage_event.synchronize()
                        fused_mlp.SYNC_WAIT[0] += time.perf_counter() - t0
                else:
                    side.wait_stream(cur)
                st.side_saw = ver
            with torch.cuda.stream(side), torch.no_grad():
                from . import pointnet2_utils
                st.inds_ring[slot].copy_(pointnet2_utils.furthest_point_sample(st.next_clouds[..., 0:3].contiguous(),
                                                                              self.prefetch.npoint))
            return slot
        slot = None
        if announced and _SAMPLE_AT == "start":
            slot = sample_next()
        g.fwd.replay()
        if announced and _SAMPLE_AT != "start":
            slot = sample_next()                         # beside the backward: see _SAMPLE_AT
        g.bwd.replay()
        # what this step enqueued behind the forward, in host order (bench.py prints it with the collective schedule: the
        # position of slice A's all-reduce relative to the second backward graph is the overlap the design claims)
        log = self.enqueue_log = ["graph: forward + loss", "graph: backward" + (" part 1 + pack slice A" if g.bwd2 is not None else
                                  " + pack" if g.update is not None else " + Adam")]
        if g.update is not None:                         # data parallel: the collectives sit between the graphs
            if g.bwd2 is not None:
                # the backward is two graphs, cut where 94 % of the gradient (everything behind level 2 of the backbone)
                # is complete and packed: that slice's all-reduce runs on the side stream BESIDE the second graph.  The
                # dependency "slice packed -> collective" is kept on the HOST (the event is waited for here, with the
                # second graph already enqueued, so the GPU does not idle): a wait packet parked in the side queue while
                # the main queue works is what costs 0.7 ms on this stack, a satisfied one is free.
                self._cut_event.record(cur)
                g.bwd2.replay()
                log += ["event behind part 1 (main stream)", "graph: backward part 2 + pack slice B (main stream)",
                        "host waits for the event", "all_reduce slice A (comm stream, beside part 2)",
                        "all_reduce slice B (main stream, behind part 2)", "main stream joins the comm stream"]
                if not self._cut_event.query():
                    t0 = time.perf_counter()
                    self._cut_event.synchronize()
                    fused_mlp.SYNC_WAIT[0] += time.perf_counter() - t0
                tm, self.grads.timing = self.grads.timing, False
                if tm:   # what the main stream still waits for once its own backward is through
                    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    ev[0].record(cur)
                self.grads.issue_packed(0, stream=self._comm)
                self.grads.reduce_flat()                 # the small second slice, behind the second graph
                cur.wait_stream(self._comm)              # (a join at the END of the main queue: free)
                if tm:
                    ev[1].record(cur)
                    self.grads._stall_events.append(ev)
                self.grads.timing = tm
            else:
                self.grads.reduce_flat()
                log.append("all_reduce of the whole flat gradient (main stream)")
            g.update.replay()
            log.append("graph: Adam")
        self.optimizer.count_step()
        self.scheduler.step()
        self.graph_replays += 1
        if announced:
            cur.wait_stream(self.prefetch.side)
            with torch.no_grad():   # (.data: device-side moves between static buffers; the identity tokens follow below)
                st.inds.data.copy_(st.inds_ring[slot])
                if not _is_self(st.next_src):
                    st.batch['point_clouds'].data.copy_(st.next_clouds)
            st.ring_done(slot, cur)
            st.moved_next_in()   # the announced clouds and their samples now sit in the current slots
        return g.loss

    def _capture(self, st, announced):
        """One eager step on the static buffers (everything lazy gets built: caches, workspaces, the allocator's pool),
        model / optimizer state restored, then the same code under stream capture."""
        dev = self.device
        opt = self.optimizer
        opt.tensor_lr = True
        opt.set_lr_tensor()
        # (size-pattern constants the label matching caches - row ids, padded buffers - must outlive the graph that baked
        # their addresses in: every entry touched from here to the end of the capture is pinned)
        from .label_generation import pinning
        with pinning():
            return self._capture_pinned(st, announced)

    def _capture_pinned(self, st, announced):
        dev = self.device
        opt = self.optimizer
        keep = {k: v.detach().clone() for k, v in self.net.state_dict().items()}
        keep_opt = (opt._exp_avg.clone(), opt._exp_avg_sq.clone(), opt._step_t.clone(), opt._steps)
        tick = fused_mlp._TRAIN_TICK[0]
        torch.cuda.synchronize(dev)
        cs = self._cstream
        cs.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(cs):
            for _ in range(2):
                # (no collectives in the warm-up steps, whatever the world size: what they compute is thrown away below,
                # and another rank may be REPLAYING this step - the collectives every rank issues per step must be the
                # schedule's, nothing else)
                self._body(st, announced, part="all")
                self.grads.zero_grad()
        torch.cuda.current_stream(dev).wait_stream(cs)
        torch.cuda.synchronize(dev)

        def restore():
            with torch.no_grad():
                sd = self.net.state_dict()
                for k, v in keep.items():
                    sd[k].copy_(v)
                opt._exp_avg.copy_(keep_opt[0]); opt._exp_avg_sq.copy_(keep_opt[1]); opt._step_t.copy_(keep_opt[2])
                opt._steps = keep_opt[3]
            fused_mlp._TRAIN_TICK[0] = tick + 1
        restore()
        if self._pool is None:
            self._pool = torch.cuda.graph_pool_handle()
        g = _StepGraph()
        g.fwd, g.bwd = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        torch.cuda.synchronize(dev)
        # a process group's watchdog thread polls its events (hipEventQuery) at any time: under the default "global"
        # capture mode that call from ANOTHER thread is an error that takes the process down; "thread_local" restricts
        # only the capturing thread (launches of autograd's worker thread into the capturing stream are captured as ever)
        import torch.distributed as dist
        mode = "thread_local" if dist.is_available() and dist.is_initialized() else "global"
        # forward (+ loss) and backward (+ update) are two graphs replayed back to back: the sampling of the next batch is
        # launched between them (_graph_step); with several ranks the update is a third, behind the all-reduce
        with torch.cuda.graph(g.fwd, pool=self._pool, stream=cs, capture_error_mode=mode):
            loss = self._body(st, announced, part="fwd")
            g.loss = loss[0].detach()
        with torch.cuda.graph(g.bwd, pool=self._pool, stream=cs, capture_error_mode=mode):
            self._body(st, announced, part="bwd_pack" if self.distributed else "bwd_step", loss=loss)
        if self.distributed:
            if loss[1] is not None:   # the rest of the backward, behind the gradient cut
                g.bwd2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g.bwd2, pool=self._pool, stream=cs, capture_error_mode=mode):
                    self._body(st, announced, part="bwd_pack2", loss=loss)
            g.update = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g.update, pool=self._pool, stream=cs, capture_error_mode=mode):
                self._body(st, announced, part="update")
        del loss
        # (capture launches nothing, but the python side effects of the step ran: undo them)
        self.grads.zero_grad()
        restore()
        return g

    def _body(self, st, announced, part, loss=None):
        """The step on the static buffers.  part: "all" (eager warm-up: everything, no collectives) | "fwd" (forward +
        loss: returns (loss with its autograd graph, gradient cut or None)) | "bwd_step" (backward + update) |
        "bwd_pack" (backward - up to the gradient cut if there is one - and its gradients packed into the flat buffer:
        the all-reduce follows outside) | "bwd_pack2" (the backward behind the cut, packed) | "update" (Adam on the
        flat buffer).  `loss` = what "fwd" returned."""
        if part == "update":
            self.optimizer.step(packed=True)
            return None
        cut_index = self.grads.cut
        if part in ("all", "fwd"):
            from .drp import GRAD_CUT, GRAD_CUT_IN, GRAD_CUT_OUT
            from .prefetch import KEY
            fused_mlp.begin_step(self.device)
            inputs = dict(st.batch)
            if self.lean_labels:
                inputs[LEAN] = True
            if st.tables:
                from .label_generation import GEOMETRY, TABLES
                inputs[TABLES] = st.tables   # the label kernels read the tensors' addresses from here (see _StaticBatch)
                if st.geometry is not None:
                    inputs[GEOMETRY] = st.geometry   # poses / grasp points at capacity: shapes independent of the batch
            if announced:
                inputs[KEY] = st.inds        # the current batch's samples: sampled one step ahead (or inline by the caller)
            if self.distributed and cut_index is not None:
                inputs[GRAD_CUT] = True      # (drp.DRP.forward: the deep part consumes a detached alias of the cut tensor)
            side_by_side = announced and _SAMPLE_AT == "start"
            if side_by_side:
                # the next batch's sampling runs beside this forward: with GB_RESERVE_CUS=1 the persistent GEMM grids
                # leave its CUs alone (GbGemmOpts.reserved_cus, baked into the captured launches) through the second
                # set-abstraction level
                fused_mlp.set_reserved_cus(_reserve(st.next_clouds.shape[0]))
                inputs['_after_level'] = {2: lambda: fused_mlp.set_reserved_cus(0)}
            try:
                end_points = self.net(inputs)
                cut = (end_points[GRAD_CUT_OUT], end_points[GRAD_CUT_IN]) if GRAD_CUT_OUT in end_points else None
                loss, end_points = get_loss(end_points)
            finally:
                fused_mlp.set_reserved_cus(0)
            loss = (loss, cut)
            if part == "fwd":
                return loss
        # ... backward (+ update).  The sampling launched beside it occupies one CU per cloud for ~1.5 of the backward's
        # ~11 ms.  Sizing the persistent GEMM grids of the whole captured backward for the CUs it leaves
        # (GB_RESERVE_CUS=1 -> GbGemmOpts.reserved_cus) measured 0.1 ms SLOWER than letting the few workgroups that find
        # their CU taken wait (same box, alternating: 17.85 vs 17.75 ms), so nothing is reserved by default.
        loss_t, cut = loss
        self.grads.hold = True   # no collective inside a capture or a warm-up: they run between the graphs
        if announced and _SAMPLE_AT != "start" and part != "all":
            fused_mlp.set_reserved_cus(_reserve(st.next_clouds.shape[0]))
        try:
            # the few-row weight gradients of the backward passes below leave as grouped launches when the block ends
            # (fused_mlp.WgradQueue): before anything packs, reduces or applies them
            with self._wgrad_queue(hooks_live=False):
                if part != "bwd_pack2":
                    loss_t.backward()
                if cut is not None and part in ("all", "bwd_step", "bwd_pack2"):
                    cut[0].backward(cut[1].grad)      # the part of the network in front of the gradient cut
        finally:
            fused_mlp.set_reserved_cus(0)
            self.grads.hold = False
        with torch.no_grad():
            if part == "bwd_pack":
                self.optimizer.pack(cut_index if cut is not None else 0, None)
            elif part == "bwd_pack2":
                self.optimizer.pack(0, cut_index)
            else:
                self.optimizer.step()
        return loss_t.detach()


def _reserve(clouds):
    """CUs the persistent GEMM grids leave to the sampling kernel beside them (one workgroup per cloud)."""
    return min(clouds, 128) if os.environ.get("GB_RESERVE_CUS", "0") != "0" else 0


class _StepGraph:
    __slots__ = ("fwd", "bwd", "bwd2", "update", "loss")

    def __init__(self):
        self.fwd = self.bwd = self.bwd2 = self.update = self.loss = None


def _leaves(obj, path=()):
    """(path, tensor) of every tensor in a collated batch (dict of tensors / nested lists of tensors)."""
    if torch.is_tensor(obj):
        yield path, obj
    elif isinstance(obj, dict):
        for k in obj:
            yield from _leaves(obj[k], path + (k,))
    elif isinstance(obj, (list, tuple)):
        for i, v in enumerate(obj):
            yield from _leaves(v, path + (i,))


def _signature(batch, skip=()):
    """Shapes / dtypes of every tensor of the batch (the keys in `skip` left out: label lists held at capacity)."""
    return tuple((p, tuple(t.shape), t.dtype) for p, t in _leaves(batch)
                 if p and not str(p[0]).startswith('_') and p[0] not in skip)


TABLE_KEYS = ('grasp_labels_list', 'grasp_offsets_list', 'grasp_tolerance_list')


class _Token:
    """Which source tensor a static buffer currently mirrors: the tensor OBJECT (a weak reference: the batch is not kept
    alive for this) and its version counter.  Not its address - a fresh batch that reuses a freed allocator block has the
    same address, shape and version 0 as the one it replaced (the copy was skipped and the graph trained
    on the previous clouds)."""
    __slots__ = ("ref", "version")

    def __init__(self, t):
        self.ref, self.version = weakref.ref(t), t._version

    def matches(self, t):
        return self.ref() is t and self.version == t._version


def _token(t):
    return _Token(t)


def _is_self(src):
    """_StaticBatch.next_src: ("self", id) when the announced clouds ARE the static current ones, else a _Token."""
    return isinstance(src, tuple) and src[0] == "self"


class _StaticBatch:
    """Device buffers with the shapes of one batch, at fixed addresses: what the captured step reads.  `batch` has the
    structure of the batch it was built from; `next_clouds` / `inds` / `inds_ring` serve the sampling prefetch (the step
    samples next_clouds on the side stream into a ring slot and ends by moving next_clouds -> batch['point_clouds'],
    the slot -> inds).
    Which source tensor every buffer currently mirrors is tracked by (address, shape, version) tokens, so a loop that keeps
    passing the same resident tensors - or the static buffers themselves - pays for no copy."""

    def __init__(self, batch, npoint, by_reference=(), capacity=None):
        self.signature = _signature(batch)
        # by_reference: list keys whose (large) tensors are NOT copied - the step reads them through device-side pointer
        # tables (label_generation.TABLES) that load() rewrites: the grasp label / offset / tolerance tensors, 2.8 GB per
        # batch at B = 4, staged as 1 KB of addresses
        self.by_reference = tuple(k for k in by_reference if k in batch)
        # capacity = (object slots per cloud, grasp point slots per object): the batch's label lists are then held at
        # CAPACITY (label_generation.LabelGeometry) - one set of buffers, one captured graph for every batch that fits,
        # whatever its objects and grasp points number (the reference's loader: both vary from scene to scene)
        self.geometry = None
        self._geo_tokens = None
        if capacity is not None:
            from .label_generation import LIST_KEYS, LabelGeometry
            self.by_reference = tuple(k for k in LIST_KEYS if k in batch)
            self.geometry = LabelGeometry(len(batch['grasp_points_list']), capacity[0], capacity[1],
                                          batch['point_clouds'].device)
            self.geometry.vad = tuple(batch['grasp_labels_list'][0][0].shape[1:4])

        def clone(obj, top=None):
            if top in self.by_reference:
                return [list(per) for per in obj]
            if torch.is_tensor(obj):
                return obj.detach().clone().contiguous()
            if isinstance(obj, dict):
                return {k: clone(v, k) for k, v in obj.items() if not str(k).startswith('_')}
            if isinstance(obj, (list, tuple)):
                return [clone(v) for v in obj]
            return obj
        self.batch = clone(batch)
        self._dst = {p: t for p, t in _leaves(self.batch) if p[0] not in self.by_reference}
        self.tables, self._table_addr, self._table_host, self._table_turn = {}, {}, {}, 0
        dev = batch['point_clouds'].device
        for k in self.by_reference:
            if self.geometry is not None and k not in TABLE_KEYS:
                continue    # (poses and grasp points are staged into the geometry's buffers, not read through tables)
            n = sum(len(per) for per in batch[k]) if self.geometry is None else self.geometry.B * self.geometry.kc
            self.tables[k] = torch.zeros(n, dtype=torch.int64, device=dev)
            # (the host may run three steps ahead of the GPU: a pinned buffer is rewritten only four loads later)
            self._table_host[k] = [torch.zeros(n, dtype=torch.int64).pin_memory() for _ in range(4)]
            self._table_addr[k] = None
        self._table_events = [None] * 4    # per ring slot: recorded behind the slot's copies to the device
        self._table_keep = [None] * 4      # per ring slot: the tensors its addresses point at
        self._load_tables(batch)
        self.loaded = {p: _token(t) for p, t in _leaves(batch) if p in self._dst}   # source each buffer mirrors
        clouds = self.batch['point_clouds']
        self._next_buf = self.next_clouds_src = self.inds = None
        self.next_src = None       # source next_clouds mirrors
        self.samp_for = None       # cloud_id() the samples in `inds` belong to
        if npoint:
            self._next_buf = torch.empty_like(clouds)
            self.next_clouds_src = self._next_buf
            self.inds = torch.zeros((clouds.shape[0], npoint), dtype=torch.int32, device=clouds.device)
            # the side stream's sampling results rotate through three slots (see Trainer._graph_step.sample_next)
            self.inds_ring = [torch.zeros_like(self.inds) for _ in range(3)]
            self.ring_events = [None, None, None]
            self.ring_pos = 0
            self.side_saw = None       # version of the static clouds the side stream is known to be ordered behind
            self.stage_event = torch.cuda.Event()

    def ring_next(self):
        """The ring slot of this step's sampling; blocks the HOST until the step that last read the slot (three steps
        ago) has finished - it practically always has."""
        slot = self.ring_pos % 3
        self.ring_pos += 1
        ev = self.ring_events[slot]
        if ev is not None and not ev.query():
            t0 = time.perf_counter()
            ev.synchronize()     # the host is three steps ahead of the GPU: waiting here is throttling, not work
            fused_mlp.SYNC_WAIT[0] += time.perf_counter() - t0
        return slot

    def ring_done(self, slot, stream):
        ev = self.ring_events[slot]
        if ev is None:
            ev = self.ring_events[slot] = torch.cuda.Event()
        ev.record(stream)

    def cloud_id(self):
        """Identity of what batch['point_clouds'] holds: the source it mirrors + its own version (in-place writes)."""
        return (self.loaded[('point_clouds',)], self.batch['point_clouds']._version)

    def _load_tables(self, batch):
        """Rewrite the device-side pointer tables for `batch`'s by-reference tensors.  The addresses travel through a ring
        of pinned host buffers; a slot is rewritten only after ITS copy to the device has executed (an event recorded
        behind the copy and waited for on the host - which also bounds how far the host runs ahead), and every slot keeps
        the tensors its addresses point at alive until it is reused, i.e. until the steps that read them are through
        (the host could overwrite a slot whose copy was still pending, and only the latest batch's label
        tensors were referenced)."""
        slot = self._table_turn % len(self._table_events)
        changed = False
        if self.geometry is not None:
            self._load_geometry(batch)
        for k in self.by_reference:
            if k not in self.tables:
                self.batch[k] = [list(per) for per in batch[k]]
                continue
            ts = [t for per in batch[k] for t in per]
            addr = tuple(t.data_ptr() for t in ts)
            if self.geometry is not None:   # slot b*kc + j; unused slots hold a valid address that nobody follows
                at = [addr[0]] * self.tables[k].numel()
                for a, sl in zip(addr, self.geometry.slots(batch)):
                    at[sl] = a
                ok = all(t.is_contiguous() and t.dtype == torch.float32 and a % 16 == 0 for t, a in zip(ts, addr))
                addr = tuple(at)
            else:
                ok = all(t.is_contiguous() and t.dtype == torch.float32 and a % 16 == 0 for t, a in zip(ts, addr))
            if addr != self._table_addr[k]:
                assert len(addr) == self.tables[k].numel() and ok, k
                if not changed:
                    ev = self._table_events[slot]
                    if ev is not None and not ev.query():
                        t0 = time.perf_counter()
                        ev.synchronize()
                        fused_mlp.SYNC_WAIT[0] += time.perf_counter() - t0
                    changed = True
                host = self._table_host[k][slot]
                host.copy_(torch.tensor(addr, dtype=torch.int64))
                self.tables[k].copy_(host, non_blocking=True)
                self._table_addr[k] = addr
            self.batch[k] = [list(per) for per in batch[k]]
        if changed:
            if self._table_events[slot] is None:
                self._table_events[slot] = torch.cuda.Event()
            self._table_events[slot].record()
            # (the tensors whose addresses went out with this slot stay referenced until the slot comes round again)
            self._table_keep[slot] = [self.batch[k] for k in self.by_reference]
            self._table_turn += 1

    def _load_geometry(self, batch):
        """Stage the poses and grasp points into the capacity buffers unless they already mirror these very tensors."""
        src = [t for k in ('object_poses_list', 'grasp_points_list') for per in batch[k] for t in per]
        toks = self._geo_tokens
        if toks is not None and len(toks) == len(src) and all(tok.matches(t) for tok, t in zip(toks, src)):
            return
        self.geometry.load(batch)
        self._geo_tokens = [_token(t) for t in src]

    def load(self, batch):
        """Make the static buffers hold `batch`; tensors that ARE the static ones, or that a buffer already mirrors, are
        not copied."""
        if batch is self.batch:
            return
        self._load_tables(batch)
        with torch.no_grad():
            for p, t in _leaves(batch):
                d = self._dst.get(p)
                if d is None or d.data_ptr() == t.data_ptr():
                    continue
                tok = self.loaded.get(p)
                if tok is None or not tok.matches(t):
                    d.copy_(t, non_blocking=True)
                    self.loaded[p] = _token(t)

    def load_next(self, clouds):
        """Stage the announced clouds.  When they ARE the static current clouds (a loop on one resident batch) nothing
        is copied: the sampling reads that buffer directly - nobody writes it."""
        cur = self.batch['point_clouds']
        if clouds.data_ptr() == cur.data_ptr():
            self.next_clouds_src = cur
            self.next_src = ("self", self.cloud_id())
            return
        self.next_clouds_src = self._next_buf
        if not (isinstance(self.next_src, _Token) and self.next_src.matches(clouds)):
            with torch.no_grad():
                self._next_buf.copy_(clouds, non_blocking=True)
            self.next_src = _token(clouds)

    @property
    def next_clouds(self):
        return self.next_clouds_src

    def moved_next_in(self):
        """Bookkeeping after a replay that announced a next batch: batch['point_clouds'] now holds next_clouds' content
        and `inds` its samples (device-side copies at the end of the graph: no version counter moved)."""
        if not _is_self(self.next_src):     # ("self", ...): the same content as before
            self.loaded[('point_clouds',)] = self.next_src
        self.samp_for = self.cloud_id()
        if _is_self(self.next_src):
            self.next_src = ("self", self.cloud_id())
