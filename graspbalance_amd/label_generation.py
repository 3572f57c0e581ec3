"""Training-time label matching with the semantics of the reference's label_generation.py
(process_grasp_labels :18-126, match_grasp_view_and_label :129-157): object-frame grasp labels are
moved into the camera frame, template views are re-assigned by 1-NN, and every seed takes the labels
of its nearest grasp point.  The two nearest-neighbour searches run on the HIP ``gb_knn1`` kernel
(``knn_modules.myknn``, 1-based indices like KNN._C).
"""
import torch

from .knn_modules import myknn
from .loss_utils import (GRASP_MAX_WIDTH, batch_viewpoint_params_to_matrix, generate_grasp_views,
                         grasp_view_rotations_on, grasp_views_on, transform_point_cloud)


def _nearest(ref_points, query_points):
    """Index (0-based) of the nearest row of ref_points (R,3) for each row of query_points (Q,3)."""
    ref = ref_points.transpose(0, 1).contiguous().unsqueeze(0)
    query = query_points.transpose(0, 1).contiguous().unsqueeze(0)
    return myknn(ref, query, k=1).view(-1) - 1


class _ShapeCache:
    """Small device constants that depend on a batch's size pattern only (row ids, index templates, padded buffers).
    Bounded - a loader whose label sizes keep changing must not grow it without end - but an entry that a CAPTURED step
    used is never evicted: the HIP graph has its address baked in, and a later replay would read whatever the allocator
    put there since.  ``pinning()`` is held by train.Trainer around its warm-up steps and captures:
    every entry created or read inside is pinned for the life of the process."""
    _pin_depth = 0

    def __init__(self, limit):
        self.limit, self.entries, self.pinned = limit, {}, set()

    def get(self, key, make):
        ent = self.entries.get(key)
        if ent is None:
            if len(self.entries) - len(self.pinned) >= self.limit:
                for k in [k for k in self.entries if k not in self.pinned]:
                    del self.entries[k]
            ent = self.entries[key] = make()
        if _ShapeCache._pin_depth:
            self.pinned.add(key)
        return ent

    @staticmethod
    @__import__("contextlib").contextmanager
    def pinning():
        _ShapeCache._pin_depth += 1
        try:
            yield
        finally:
            _ShapeCache._pin_depth -= 1


_ROW_ID_CACHE = _ShapeCache(256)
_ONES_CACHE = _ShapeCache(256)
_PAD_CACHE = _ShapeCache(64)
pinning = _ShapeCache.pinning


def _row_ids(sizes, first_obj, device):
    def make():
        oid = torch.cat([torch.full((n,), first_obj + k, dtype=torch.int32) for k, n in enumerate(sizes)])
        loc = torch.cat([torch.arange(n, dtype=torch.int32) for n in sizes])
        return oid.to(device), loc.to(device)
    return _ROW_ID_CACHE.get((sizes, first_obj, str(device)), make)


def _row_ids_all(sizes, device):
    """_row_ids of every cloud of the batch, concatenated (object ids count on through the batch)."""
    def make():
        oids, locs, k0 = [], [], 0
        for sz in sizes:
            oid, loc = _row_ids(sz, k0, device)
            oids.append(oid)
            locs.append(loc)
            k0 += len(sz)
        return torch.cat(oids), torch.cat(locs)
    return _ROW_ID_CACHE.get(("all", sizes, str(device)), make)


def _transform_3x4(cloud, pose):
    """transform_point_cloud(cloud, pose, '3x4') with the column of ones taken from a per-size cache (same matmul,
    same values; saves the ones() launch per object)."""
    ones = _ONES_CACHE.get((int(cloud.size(0)), cloud.dtype, str(cloud.device)), lambda: cloud.new_ones(cloud.size(0), 1))
    homo = torch.cat([cloud, ones], dim=1)
    return torch.matmul(pose, homo.T).T[:, :3]


def _transform_all(gps_all, all_poses):
    """``transform_point_cloud(gp, pose, '3x4')`` of EVERY object of the batch as one batched matmul instead of a
    cat + matmul + slice per object: the clouds are packed (one cat), scattered into a cached (K, nmax, 4) homogeneous
    buffer through a cached index (it depends on the sizes only), multiplied by their poses in one bmm, and gathered
    back.  rocBLAS forms each of the four-term dot products the same way in both shapes, so the result is the
    per-object matmul's bit for bit (asserted on the full batch in tests/test_model_gpu.py, with unequal sizes).
    Returns (T,3), objects one after another."""
    dev = all_poses.device
    sizes = tuple(int(gp.size(0)) for gp in gps_all)

    def make():
        nmax = max(sizes)
        flat = torch.cat([torch.arange(n, dtype=torch.int64) + k * nmax for k, n in enumerate(sizes)]).to(dev)
        buf = torch.ones((len(sizes) * nmax, 4), dtype=all_poses.dtype, device=dev)  # column 3 stays 1
        return flat, buf, nmax
    flat, buf, nmax = _PAD_CACHE.get((sizes, str(dev), all_poses.dtype), make)
    K = len(sizes)
    buf[:, :3].index_copy_(0, flat, torch.cat(gps_all, 0))  # padding rows keep stale values: never read back
    out = torch.bmm(all_poses, buf.view(K, nmax, 4).transpose(1, 2))  # (K,3,nmax) == matmul(pose, homo.T) per object
    return out.transpose(1, 2).reshape(K * nmax, 3).index_select(0, flat)


def _assign_views(poses, V):
    """For every object pose (K,3,4): index of the transformed template view nearest to each template
    view, (K,V) — the reference's per-object 300x300 kNN (label_generation.py:56-58), as ONE batched
    launch over all objects of the batch."""
    views = grasp_views_on(poses.device, V)                              # (V,3)
    trans = torch.matmul(poses[:, :3, :3], views.T)                      # (K,3,V) == (R v)^T per object
    query = views.T.contiguous().unsqueeze(0).expand(poses.size(0), -1, -1).contiguous()
    return myknn(trans.contiguous(), query, k=1).view(poses.size(0), V) - 1


def _label_gather(tensors, obj, pt, view_inds, V, W, want_max=False, col=None):
    """out[r,v,:] = tensors[obj[r]][pt[r], view_inds[obj[r], v], :] through the fused HIP gather; with want_max also
    the maximum of the gathered values (a 0-d tensor, == out.max()) from the same pass; with col = (stride, offset)
    also a contiguous copy of the columns w % stride == offset (the widths of the offsets tensor).
    The objects' label tensors are new every step in training: their addresses travel in the kernel arguments (a host
    array), not through a device-side table that would need a copy - or a cache that only a benchmark could hit."""
    import ctypes
    from . import _lib
    dev = obj.device
    out = torch.empty((obj.numel(), V, W), dtype=torch.float32, device=dev)
    out_max = torch.full((), float("-inf"), dtype=torch.float32, device=dev) if want_max else None
    out_col = torch.empty((obj.numel(), V, W // col[0]), dtype=torch.float32, device=dev) if col else None
    table = (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    with _lib.device_ctx(dev):
        _lib.check(_lib.lib().gb_label_gather(ctypes.cast(table, ctypes.c_void_p), len(tensors), _lib.ptr(obj),
                                              _lib.ptr(pt), _lib.ptr(view_inds), _lib.ptr(out), _lib.ptr(out_max),
                                              _lib.ptr(out_col), col[0] if col else 1, col[1] if col else 0,
                                              obj.numel(), V, W, _lib.current_stream(dev)), "label_gather")
    res = (out,)
    if want_max:
        res += (out_max,)
    if col:
        res += (out_col,)
    return res if len(res) > 1 else out


def _finish_labels(end_points, batch, batch_size, num_samples):
    labels = batch['label']  # (B,Ns,V,A,D)
    V, A, D = labels.shape[2:]
    offsets = batch['offset']
    u_max = batch.get('label_max')  # from the gather pass (fused path); else one more pass over the tensor
    if u_max is None:
        u_max = labels.max()
    if (labels.is_cuda and (A * D) % 4 == 0 and A * D <= 256 and labels.is_contiguous() and offsets.is_contiguous()
            and labels.dtype == torch.float32 and offsets.dtype == torch.float32):
        # one pass (gb_label_finish) instead of compare / and / clamp / div / log / where / max over (B,Ns,V,A,D)
        from . import _lib
        out = torch.empty_like(labels)
        view_scores = torch.empty((batch_size, num_samples, V), dtype=torch.float32, device=labels.device)
        view_arg = torch.empty((batch_size, num_samples, V), dtype=torch.int32, device=labels.device)
        end_points['_view_label_arg'] = view_arg  # where in (A,D) each view's maximum sits (first one)
        with _lib.device_ctx(labels.device):
            _lib.check(_lib.lib().gb_label_finish(_lib.ptr(labels), _lib.ptr(offsets), _lib.ptr(batch.get('width')),
                                                  _lib.ptr(u_max), float(GRASP_MAX_WIDTH), _lib.ptr(out),
                                                  _lib.ptr(view_scores),
                                                  _lib.ptr(view_arg),
                                                  batch_size * num_samples * V, A * D,
                                                  _lib.current_stream(labels.device)), "gb_label_finish")
        labels = out
    else:
        widths = offsets[:, :, :, :, :, 2]
        label_mask = (labels > 0) & (widths <= GRASP_MAX_WIDTH)
        # == labels[mask] = log(u_max / labels[mask]); labels[~mask] = 0   (no boolean-index host sync)
        labels = torch.where(label_mask, torch.log(u_max / labels.clamp_min(1e-30)), torch.zeros_like(labels))
        view_scores, _ = labels.view(batch_size, num_samples, V, A * D).max(dim=-1)
    end_points['batch_grasp_point'] = batch['point']
    end_points['batch_grasp_view'] = batch['view']
    end_points['batch_grasp_view_rot'] = batch['view_rot']
    end_points['batch_grasp_label'] = labels
    end_points['batch_grasp_offset'] = batch['offset']
    end_points['batch_grasp_tolerance'] = batch['tolerance']
    end_points['batch_grasp_view_label'] = view_scores.float()
    end_points['_view_label_source'] = labels  # lets the loss reuse the per-view maximum (same tensor object)
    return end_points


def _process_grasp_labels_fused(end_points):
    """Same results as the per-object composition below with the two index maps (view permutation, then
    nearest grasp point per seed) composed: only the (B,Ns) rows that are kept are ever copied."""
    seed_xyzs = end_points['fp2_xyz']
    B, Ns, _ = seed_xyzs.shape
    dev = seed_xyzs.device
    poses_l = end_points['object_poses_list']
    labels_l = [t for per in end_points['grasp_labels_list'] for t in per]
    offsets_l = [t for per in end_points['grasp_offsets_list'] for t in per]
    tol_l = [t for per in end_points['grasp_tolerance_list'] for t in per]
    all_poses = torch.stack([p for poses in poses_l for p in poses], 0)          # (Kt,3,4)
    _, V, A, D = labels_l[0].shape
    view_inds = _assign_views(all_poses, V).contiguous()                         # (Kt,V) int64
    views = grasp_views_on(dev, V)
    rot_template = grasp_view_rotations_on(dev, V)   # == batch_viewpoint_params_to_matrix(-views, 0): a constant
    R = all_poses[:, :3, :3]
    views_trans = torch.matmul(R, views.T).transpose(1, 2)                       # (Kt,V,3)
    # R_k @ T_v for every (object, view) as ONE (3 Kt x 3) @ (3 x 3 V) product instead of a 3x3 bmm with Kt*V batches
    # (94 us): the same three-term dot products - bit-identical, tests/test_model_gpu.py compares the rotations exactly
    Kt = R.size(0)
    rot_trans = torch.mm(R.reshape(Kt * 3, 3), rot_template.permute(1, 0, 2).reshape(3, V * 3)) \
        .view(Kt, 3, V, 3).permute(0, 2, 1, 3)                                   # (Kt,V,3,3), strided
    views_sel = torch.gather(views_trans, 1, view_inds.unsqueeze(-1).expand(-1, -1, 3))
    rot_sel = torch.gather(rot_trans, 1, view_inds.view(-1, V, 1, 1).expand(-1, -1, 3, 3))
    gps_l = end_points['grasp_points_list']
    pts_all = _transform_all([gp for gps in gps_l for gp in gps], all_poses)  # every object's grasp points, camera frame
    sizes = tuple(tuple(int(gp.size(0)) for gp in gps) for gps in gps_l)
    per_cloud = [sum(sz) for sz in sizes]
    if len(set(per_cloud)) == 1 and per_cloud[0] > 0:
        # every cloud brings the same number of grasp points (always true for a fixed number of points per object): the B
        # nearest-grasp-point searches are ONE batched kNN launch and the three row gathers one each - the same values
        # as the per-cloud calls below (the kernel treats batch elements independently; lowest index among equals)
        n = per_cloud[0]
        ref = pts_all.view(B, n, 3).transpose(1, 2).contiguous()
        query = seed_xyzs.transpose(1, 2).contiguous()
        nn_inds = myknn(ref, query, k=1).view(B, Ns) - 1
        flat = (nn_inds + torch.arange(B, device=dev, dtype=nn_inds.dtype).view(B, 1) * n).view(-1)
        oid, loc = _row_ids_all(sizes, dev)
        points = torch.index_select(pts_all, 0, flat).view(B, Ns, 3)
        obj = torch.index_select(oid, 0, flat)
        pt = torch.index_select(loc, 0, flat)
    else:
        obj_of_seed, pt_of_seed, points = [], [], []
        k0 = 0
        p0 = 0
        for i in range(B):
            gps = gps_l[i]
            n_i = per_cloud[i]
            pts = pts_all[p0:p0 + n_i]  # == cat([transform_point_cloud(gp, pose, '3x4') for the cloud's objects])
            p0 += n_i
            # (object id, point id within the object) of every row of pts: depends on the sizes only - built once per
            # size pattern instead of a full_() + arange() pair per object per step
            oid, loc = _row_ids(sizes[i], k0, dev)
            k0 += len(poses_l[i])
            nn_inds = _nearest(pts, seed_xyzs[i])
            points.append(torch.index_select(pts, 0, nn_inds))
            obj_of_seed.append(torch.index_select(oid, 0, nn_inds))
            pt_of_seed.append(torch.index_select(loc, 0, nn_inds))
        obj = torch.cat(obj_of_seed, 0).contiguous()
        pt = torch.cat(pt_of_seed, 0).contiguous()
        points = torch.stack(points, 0)
    objl = obj.long()
    if end_points.get(LEAN) and 'grasp_top_view_inds' in end_points:
        return _lean_labels(end_points, labels_l, offsets_l, tol_l, obj, pt, view_inds, views_sel, rot_sel, objl,
                            points, B, Ns, V, A, D)
    label, label_max = _label_gather(labels_l, obj, pt, view_inds, V, A * D, want_max=True)
    offset, width = _label_gather(offsets_l, obj, pt, view_inds, V, A * D * 3, col=(3, 2))
    batch = {
        'point': points,
        'view': torch.index_select(views_sel, 0, objl).view(B, Ns, V, 3),
        'view_rot': torch.index_select(rot_sel, 0, objl).view(B, Ns, V, 3, 3),
        'label': label.view(B, Ns, V, A, D),
        'label_max': label_max,
        'offset': offset.view(B, Ns, V, A, D, 3),
        'width': width,  # offsets[..., 2], contiguous: what the score transform reads
        'tolerance': _label_gather(tol_l, obj, pt, view_inds, V, A * D).view(B, Ns, V, A, D),
    }
    return _finish_labels(end_points, batch, B, Ns)


# ---- capacity form (round 5; ADVICE round 4: a captured step keyed on every label tensor's shape is useless on the
# reference's loader, whose per-scene object counts and per-object grasp point counts vary) ---------------------------
GEOMETRY = '_label_geometry'   # end_points entry (train._StaticBatch): LabelGeometry - the batch's objects at CAPACITY
LIST_KEYS = ('object_poses_list', 'grasp_points_list', 'grasp_labels_list', 'grasp_offsets_list', 'grasp_tolerance_list')
FAR = 1.0e18                   # coordinate of a padding grasp point: never anybody's nearest neighbour (its square is finite)


def label_needs(batch):
    """(objects per cloud, grasp points per object) the batch needs at most."""
    return (max(len(per) for per in batch['grasp_points_list']),
            max(int(gp.shape[0]) for per in batch['grasp_points_list'] for gp in per))


class LabelGeometry:
    """The object poses and grasp points of a batch in buffers of FIXED shape: `kc` object slots per cloud and `pc` point
    slots per object, whatever the batch brings (poses (B*kc,3,4), unused slots the identity; points (B*kc*pc,4)
    homogeneous, unused slots FAR away).  Label matching on it has the same launches, grids and addresses for every
    batch that fits, so ONE captured step serves a loader whose sizes change from scene to scene; the per-object label
    tensors themselves are read through the device-side pointer tables (slot b*kc + j)."""

    def __init__(self, B, kc, pc, device):
        self.B, self.kc, self.pc = B, kc, pc
        self.poses = torch.zeros((B * kc, 3, 4), dtype=torch.float32, device=device)
        self.points = torch.empty((B * kc * pc, 4), dtype=torch.float32, device=device)
        self._eye = torch.eye(3, 4, dtype=torch.float32, device=device)
        self._pattern = None

    def fits(self, batch):
        k, n = label_needs(batch)
        return len(batch['grasp_points_list']) == self.B and k <= self.kc and n <= self.pc

    def slots(self, batch):
        """Object slot (b*kc + j) of every object of the batch, in list order."""
        return [b * self.kc + j for b, per in enumerate(batch['grasp_points_list']) for j in range(len(per))]

    def load(self, batch):
        """Stage the batch's poses and grasp points (eager copies, outside any graph)."""
        gps = [gp for per in batch['grasp_points_list'] for gp in per]
        poses = [p for per in batch['object_poses_list'] for p in per]
        dev = self.poses.device
        sizes = tuple(int(gp.shape[0]) for gp in gps)
        slots = self.slots(batch)
        pattern = (sizes, tuple(slots))
        if pattern != self._pattern:   # where every packed row goes: depends on the size pattern only
            rows = torch.cat([torch.arange(n, dtype=torch.int64) + sl * self.pc for n, sl in zip(sizes, slots)])
            self._rows, self._slots = rows.to(dev), torch.tensor(slots, dtype=torch.int64, device=dev)
            self._pattern = pattern
        with torch.no_grad():
            self.points[:, :3] = FAR
            self.points[:, 3] = 1.0
            self.points[:, :3].index_copy_(0, self._rows, torch.cat(gps, 0).to(torch.float32))
            self.poses.copy_(self._eye.expand_as(self.poses))
            self.poses.index_copy_(0, self._slots, torch.stack(poses, 0).to(torch.float32))


def _match_at_capacity(geo, seed_xyzs):
    """(points (B,Ns,3), obj (B*Ns) int32 object slot, pt (B*Ns) int32 point within the object) of every seed's nearest
    grasp point - the reference's per-cloud kNN over the cloud's transformed grasp points (label_generation.py:84) on
    the padded buffers.  Valid points keep their relative order inside a cloud (objects, then points), so "lowest index
    among equally near" picks the same point as the packed search."""
    B, Ns, _ = seed_xyzs.shape
    Kt, kc, pc = geo.poses.shape[0], geo.kc, geo.pc
    pts = torch.bmm(geo.poses, geo.points.view(Kt, pc, 4).transpose(1, 2)).transpose(1, 2).reshape(Kt * pc, 3)
    ref = pts.view(B, kc * pc, 3).transpose(1, 2).contiguous()
    query = seed_xyzs.transpose(1, 2).contiguous()
    nn_inds = myknn(ref, query, k=1).view(B, Ns) - 1
    flat = (nn_inds + torch.arange(B, device=seed_xyzs.device, dtype=nn_inds.dtype).view(B, 1) * (kc * pc)).view(-1)
    points = torch.index_select(pts, 0, flat).view(B, Ns, 3)
    obj = torch.div(flat, pc, rounding_mode='floor').to(torch.int32)
    pt = (flat - obj.long() * pc).to(torch.int32)
    return points, obj.contiguous(), pt.contiguous()


def _process_grasp_labels_at_capacity(end_points):
    """_process_grasp_labels_fused on a LabelGeometry + device-side pointer tables: no tensor shape depends on how many
    objects or grasp points the batch has (only the lean form exists: it is what a training step runs)."""
    geo, tables = end_points[GEOMETRY], end_points[TABLES]
    seed_xyzs = end_points['fp2_xyz']
    B, Ns, _ = seed_xyzs.shape
    dev = seed_xyzs.device
    V, A, D = geo.vad
    all_poses = geo.poses
    view_inds = _assign_views(all_poses, V).contiguous()                         # (Kt,V) int64
    views = grasp_views_on(dev, V)
    rot_template = grasp_view_rotations_on(dev, V)
    R = all_poses[:, :3, :3]
    views_trans = torch.matmul(R, views.T).transpose(1, 2)                       # (Kt,V,3)
    Kt = R.size(0)
    rot_trans = torch.mm(R.reshape(Kt * 3, 3), rot_template.permute(1, 0, 2).reshape(3, V * 3)) \
        .view(Kt, 3, V, 3).permute(0, 2, 1, 3)                                   # (Kt,V,3,3), strided
    views_sel = torch.gather(views_trans, 1, view_inds.unsqueeze(-1).expand(-1, -1, 3))
    rot_sel = torch.gather(rot_trans, 1, view_inds.view(-1, V, 1, 1).expand(-1, -1, 3, 3))
    points, obj, pt = _match_at_capacity(geo, seed_xyzs)
    return _lean_labels(end_points, [None] * Kt, [None] * Kt, [None] * Kt, obj, pt, view_inds, views_sel, rot_sel,
                        obj.long(), points, B, Ns, V, A, D)


LEAN = '_lean_labels'   # end_points flag (train.Trainer sets it): build only what a training step consumes
TABLES = '_label_tables'  # end_points entry (train._StaticBatch): {list key: int64 device tensor of the tensors' addresses}


BY_REFERENCE = ('grasp_labels_list', 'grasp_offsets_list', 'grasp_tolerance_list')


def tables_ok(batch):
    """Can the lean label matching read this batch's label tensors through device-side pointer tables?  (What _fusable
    asks of them: fp32, contiguous, 16-byte aligned, at most 128 objects, one (V,A,D) for all.)"""
    if not all(k in batch for k in BY_REFERENCE):
        return False
    ts = [t for key in BY_REFERENCE for per in batch[key] for t in per]
    if not ts or len(ts) > 3 * 128:
        return False
    shape = ts[0].shape[1:4]
    return all(t.is_cuda and t.is_contiguous() and t.dtype == torch.float32 and t.data_ptr() % 16 == 0
               and t.shape[1:4] == shape for t in ts)


def _table(tensors):
    import ctypes
    return ctypes.cast((ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors]), ctypes.c_void_p)


def _lean_labels(end_points, labels_l, offsets_l, tol_l, obj, pt, view_inds, views_sel, rot_sel, objl, points, B, Ns, V,
                 A, D):
    """The training step's label matching without the (B,Ns,V,A,D[,3]) tensors (1.2 GB written and 2.3 GB moved per step
    at B = 4, of which the step reads the per-view maxima and ONE view per seed): one read pass for the labels' maximum,
    one for the view labels (gb_label_scores), then the rows of the view the network picked and of the seed's best view
    (gb_label_gather_view).  Same values as the full path for every key a train step reads
    (tests/test_model_gpu.py::test_lean_label_matching_equals_full); the *_all tensors are not produced."""
    from . import _lib
    dev = obj.device
    R, AD = obj.numel(), A * D
    lib = _lib.lib()
    st = _lib.current_stream(dev)
    tables = end_points.get(TABLES)
    if tables is not None:
        # device-side pointer tables kept by the caller (train._StaticBatch): a captured step follows their CONTENT
        lab_t, off_t, tol_t = (tables[k].data_ptr() for k in ('grasp_labels_list', 'grasp_offsets_list', 'grasp_tolerance_list'))
        gather, scores, gather_view = lib.gb_label_gather_dt, lib.gb_label_scores_dt, lib.gb_label_gather_view_dt
    else:
        lab_t, off_t, tol_t = _table(labels_l), _table(offsets_l), _table(tol_l)
        gather, scores, gather_view = lib.gb_label_gather, lib.gb_label_scores, lib.gb_label_gather_view
    u_max = torch.full((), float("-inf"), dtype=torch.float32, device=dev)
    view_scores = torch.empty((B, Ns, V), dtype=torch.float32, device=dev)
    view_arg = torch.empty((B, Ns, V), dtype=torch.int32, device=dev)
    with _lib.device_ctx(dev):
        _lib.check(gather(lab_t, len(labels_l), _lib.ptr(obj), _lib.ptr(pt), _lib.ptr(view_inds), None,
                          _lib.ptr(u_max), None, 1, 0, R, V, AD, st), "gb_label_gather (max)")
        _lib.check(scores(lab_t, off_t, len(labels_l), _lib.ptr(obj), _lib.ptr(pt), _lib.ptr(view_inds),
                                       _lib.ptr(u_max), float(GRASP_MAX_WIDTH), _lib.ptr(view_scores), _lib.ptr(view_arg),
                                       R, V, AD, st), "gb_label_scores")
        top = end_points['grasp_top_view_inds'].reshape(R).long().contiguous()
        best_view = view_scores.view(R, V).argmax(dim=1)            # first maximum: with view_arg the flat arg-max of loss.py:31
        rows = {}
        for name, tab, n, W, rv in (("label", lab_t, len(labels_l), AD, top), ("offset", off_t, len(offsets_l), AD * 3, top),
                                    ("tolerance", tol_t, len(tol_l), AD, top), ("best_offset", off_t, len(offsets_l), AD * 3, best_view)):
            out = torch.empty((R, W), dtype=torch.float32, device=dev)
            _lib.check(gather_view(tab, n, _lib.ptr(obj), _lib.ptr(pt), _lib.ptr(view_inds), _lib.ptr(rv),
                                   _lib.ptr(out), R, V, W, st), "gb_label_gather_view")
            rows[name] = out
    raw, off = rows["label"], rows["offset"].view(R, AD, 3)
    mask = (raw > 0) & (off[:, :, 2] <= GRASP_MAX_WIDTH)
    top_label = torch.where(mask, torch.log(u_max / raw.clamp_min(1e-30)), torch.zeros_like(raw))
    best_ad = torch.gather(view_arg.view(R, V), 1, best_view.unsqueeze(1)).long()                      # (R,1)
    seed_width = torch.gather(rows["best_offset"].view(R, AD, 3)[:, :, 2], 1, best_ad).view(B, Ns)
    end_points['batch_grasp_point'] = points
    end_points['batch_grasp_view'] = torch.index_select(views_sel, 0, objl).view(B, Ns, V, 3)
    end_points['batch_grasp_view_rot'] = torch.index_select(rot_sel, 0, objl).view(B, Ns, V, 3, 3)
    end_points['batch_grasp_view_label'] = view_scores
    end_points['_view_label_arg'] = view_arg
    end_points['_lean'] = {'label': top_label.view(B, Ns, A, D), 'offset': off.view(B, Ns, A, D, 3),
                           'tolerance': rows["tolerance"].view(B, Ns, A, D), 'seed_width': seed_width}
    return end_points


def _fusable(end_points):
    if not end_points['fp2_xyz'].is_cuda:
        return False
    ts = [t for key in ('grasp_labels_list', 'grasp_offsets_list', 'grasp_tolerance_list')
          for per in end_points[key] for t in per]
    shape = ts[0].shape[1:4]
    return len(ts) <= 3 * 128 and all(t.is_contiguous() and t.dtype == torch.float32 and t.data_ptr() % 16 == 0
                                        for t in ts) and all(t.shape[1:4] == shape for t in ts)  # <= 128 objects per batch


def process_grasp_labels(end_points):
    if end_points.get(GEOMETRY) is not None and end_points.get(LEAN) and 'grasp_top_view_inds' in end_points:
        return _process_grasp_labels_at_capacity(end_points)
    if _fusable(end_points):
        return _process_grasp_labels_fused(end_points)
    seed_xyzs = end_points['fp2_xyz']  # (B,Ns,3)
    batch_size, num_samples, _ = seed_xyzs.size()
    per_cloud = {k: [] for k in ('point', 'view', 'view_rot', 'label', 'offset', 'tolerance')}
    all_poses = torch.stack([p for poses in end_points['object_poses_list'] for p in poses], 0)
    V_all = end_points['grasp_labels_list'][0][0].size(1)
    all_view_inds = _assign_views(all_poses, V_all)
    flat_obj = 0
    for i in range(len(end_points['input_xyz'])):
        merged = {k: [] for k in per_cloud}
        for obj_idx, pose in enumerate(end_points['object_poses_list'][i]):
            grasp_points = end_points['grasp_points_list'][i][obj_idx]        # (Np,3)
            grasp_labels = end_points['grasp_labels_list'][i][obj_idx]        # (Np,V,A,D)
            grasp_offsets = end_points['grasp_offsets_list'][i][obj_idx]      # (Np,V,A,D,3)
            grasp_tolerance = end_points['grasp_tolerance_list'][i][obj_idx]  # (Np,V,A,D)
            _, V, A, D = grasp_labels.size()
            num_grasp_points = grasp_points.size(0)
            # template views and their rotations, moved by the object pose
            grasp_views = generate_grasp_views(V).to(pose.device)
            grasp_points_trans = transform_point_cloud(grasp_points, pose, '3x4')
            grasp_views_trans = transform_point_cloud(grasp_views, pose[:3, :3], '3x3')
            angles = torch.zeros(V, dtype=grasp_views.dtype, device=grasp_views.device)
            grasp_views_rot = batch_viewpoint_params_to_matrix(-grasp_views, angles)
            grasp_views_rot_trans = torch.matmul(pose[:3, :3], grasp_views_rot)
            # each template view takes the labels of the nearest transformed view
            view_inds = all_view_inds[flat_obj]
            flat_obj += 1
            merged['point'].append(grasp_points_trans)
            merged['view'].append(torch.index_select(grasp_views_trans, 0, view_inds)
                                  .unsqueeze(0).expand(num_grasp_points, -1, -1))
            merged['view_rot'].append(torch.index_select(grasp_views_rot_trans, 0, view_inds)
                                      .unsqueeze(0).expand(num_grasp_points, -1, -1, -1))
            merged['label'].append(torch.index_select(grasp_labels, 1, view_inds))
            merged['offset'].append(torch.index_select(grasp_offsets, 1, view_inds))
            merged['tolerance'].append(torch.index_select(grasp_tolerance, 1, view_inds))
        merged = {k: torch.cat(v, dim=0) for k, v in merged.items()}  # (Np', ...)
        nn_inds = _nearest(merged['point'], seed_xyzs[i])  # (Ns,)
        for k in per_cloud:
            per_cloud[k].append(torch.index_select(merged[k], 0, nn_inds))
    batch = {k: torch.stack(v, 0) for k, v in per_cloud.items()}
    return _finish_labels(end_points, batch, batch_size, num_samples)


def _take_view(t, top_view_inds):
    """t (B,Ns,V,...) -> (B,Ns,...) selecting view top_view_inds[b,s]."""
    B, Ns = top_view_inds.shape
    tail = t.shape[3:]
    index = top_view_inds.view(B, Ns, 1, *([1] * len(tail))).expand(B, Ns, 1, *tail)
    return torch.gather(t, 2, index).squeeze(2)


def match_grasp_view_and_label(end_points):
    top_view_inds = end_points['grasp_top_view_inds']       # (B,Ns)
    template_views_rot = end_points['batch_grasp_view_rot']  # (B,Ns,V,3,3)
    template_views = end_points['batch_grasp_view']          # (B,Ns,V,3)
    lean = end_points.get('_lean')
    if lean is not None:   # the top view's rows were gathered straight from the objects' tensors (_lean_labels)
        top_rot = _take_view(template_views_rot, top_view_inds)
        end_points['batch_grasp_view_rot'] = top_rot
        end_points['batch_grasp_view'] = _take_view(template_views, top_view_inds)
        end_points['batch_grasp_view_all'] = template_views
        end_points['batch_grasp_label'] = lean['label']
        end_points['batch_grasp_offset'] = lean['offset']
        end_points['batch_grasp_tolerance'] = lean['tolerance']
        end_points['_seed_width'] = lean['seed_width']
        return top_rot, lean['label'], lean['offset'], lean['tolerance'], end_points
    grasp_labels = end_points['batch_grasp_label']           # (B,Ns,V,A,D)
    grasp_offsets = end_points['batch_grasp_offset']         # (B,Ns,V,A,D,3)
    grasp_tolerance = end_points['batch_grasp_tolerance']    # (B,Ns,V,A,D)
    top_rot = _take_view(template_views_rot, top_view_inds)
    top_labels = _take_view(grasp_labels, top_view_inds)
    top_offsets = _take_view(grasp_offsets, top_view_inds)
    top_tolerance = _take_view(grasp_tolerance, top_view_inds)
    end_points['batch_grasp_view_rot'] = top_rot
    end_points['batch_grasp_view'] = _take_view(template_views, top_view_inds)
    end_points['batch_grasp_view_all'] = template_views
    end_points['batch_grasp_label'] = top_labels
    end_points['batch_grasp_label_all'] = grasp_labels
    end_points['batch_grasp_offset'] = top_offsets
    end_points['batch_grasp_offset_all'] = grasp_offsets
    end_points['batch_grasp_tolerance'] = top_tolerance
    return top_rot, top_labels, top_offsets, top_tolerance, end_points
