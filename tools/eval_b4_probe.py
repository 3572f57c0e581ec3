"""Probe: where do the grasp-score tensors of the full-size eval forward leave the fp64 truth at B = 4?"""
import copy, sys, torch
sys.path.insert(0, '.')
from tests import f64_truth
from tests.f64_truth import rel
from tests.seeded import fill_by_key
from tests.test_parity_f64_gpu import _force, _truth_forward
from graspbalance_amd.graspbalance import GraspBalance
from graspbalance_amd.scene import make_batch
DEV = "cuda:0"
net = fill_by_key(GraspBalance(is_training=False), seed=21).eval()
clouds = torch.from_numpy(make_batch([0, 1, 2, 3], 20000))
gpu = copy.deepcopy(net).to(DEV)
with torch.no_grad():
    free = gpu({'point_clouds': clouds.to(DEV)})
views = free['grasp_top_view_inds'].cpu()
_force(gpu, views)
with torch.no_grad():
    got = gpu({'point_clouds': clouds.to(DEV)})
    _, truth = _truth_forward(gpu, {'point_clouds': clouds.double().to(DEV)}, views)
for k in ('fp2_features', 'objectness_score', 'grasp_score_pred', 'grasp_width_pred'):
    print(k, ["%.1e" % rel(got[k][i], truth[k][i]) for i in range(4)])
g, t = got['grasp_score_pred'].double(), truth['grasp_score_pred']
err = (g - t).abs()
print("max abs err", float(err.max()), "max |t|", float(t.abs().max()), "rms t", float(t.pow(2).mean().sqrt()))
per_seed = err.amax(dim=(1, 3))          # (B, Ns)
for b in range(4):
    top = torch.topk(per_seed[b], 5)
    print("cloud", b, "worst seeds", top.indices.tolist(), ["%.1e" % v for v in top.values.tolist()],
          "median %.1e" % float(per_seed[b].median()))
# ---- where inside stage 2?
store = {}
def hook(tag):
    def f(mod, inp, out=None):
        x = inp[0]
        store[tag] = x.detach() if torch.is_tensor(x) else None
    return f
def run(net_, batch, tag):
    hs = [net_.grasp_generator.GraspParameters.register_forward_pre_hook(hook(tag + "/vp")),
          net_.grasp_generator.GraspParameters.conv3.register_forward_pre_hook(hook(tag + "/h2")),
          net_.grasp_generator.GraspParameters.conv2.register_forward_pre_hook(hook(tag + "/h1"))]
    for i in (1, 2, 3, 4):
        g = getattr(net_.grasp_generator, "WidthGroup%d" % i)
        hs.append(g.register_forward_hook(lambda m, a, o, i=i: store.__setitem__(tag + "/wg%d" % i, o.detach())))
    out = net_(batch)
    for h in hs:
        h.remove()
    return out
from graspbalance_amd import fused_mlp
with torch.no_grad():
    run(gpu, {'point_clouds': clouds.to(DEV)}, "hip")
    net64 = f64_truth.double_model(gpu)
    _force(net64, views)
    fused_mlp.set_enabled(False)
    with f64_truth.torch_geometry(), f64_truth.double_stage2_inputs():
        run(net64, {'point_clouds': clouds.double().to(DEV)}, "f64")
    fused_mlp.set_enabled(True)
B = 4
for k in ("vp", "h1", "h2"):
    a, b = store["hip/" + k], store["f64/" + k]
    a = a.reshape(B, a.shape[1], -1); b = b.reshape(B, b.shape[1], -1)
    print(k, tuple(a.shape), ["%.1e" % rel(a[i], b[i]) for i in range(B)], "rms", ["%.1e" % float(b[i].pow(2).mean().sqrt()) for i in range(B)])
for i in (1, 2, 3, 4):
    a, b = store["hip/wg%d" % i], store["f64/wg%d" % i]          # hip: rows (B*Ns*D, C); f64: (B, C, Ns, D)
    b = b.permute(0, 2, 3, 1).reshape(B, -1, b.shape[1]); a = a.reshape(B, -1, a.shape[-1])
    print("wg%d" % i, ["%.1e" % rel(a[j], b[j]) for j in range(B)], "rms", ["%.1e" % float(b[j].pow(2).mean().sqrt()) for j in range(B)])
