"""Kernel-only timing of the row-streaming products (csrc/gemm_rs.hip) through a given build of the library.
usage: rs_probe.py <lib.so> [iters] [precision]     (precision: 0 = fp32 MFMA, 2 = GB_PREC_F32_SPLIT3)"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
so = ctypes.CDLL(os.path.abspath(sys.argv[1]))
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
P_, LL, I = ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int
so.gb_gemm_fwd.argtypes = [P_, P_, P_, P_, P_, I, LL, I, I, P_, P_, P_]
so.gb_gemm_dgrad.argtypes = [P_, P_, P_, P_, P_, P_, I, LL, I, I, P_, P_, P_, P_, P_]
so.gb_gemm_fwd_pool.argtypes = [P_, P_, P_, P_, P_, P_, LL, LL, P_, P_, I, LL, I, I, I, P_, P_, P_]
dev = "cuda:0"
st = torch.cuda.current_stream().cuda_stream


class GemmOpts(ctypes.Structure):
    _fields_ = [("precision", ctypes.c_int), ("reserved_cus", ctypes.c_int), ("scratch", ctypes.c_void_p),
                ("scratch_bytes", ctypes.c_ulonglong), ("rows_dev", ctypes.c_void_p), ("flags", ctypes.c_int)]


OPTS = ctypes.pointer(GemmOpts(int(sys.argv[3]) if len(sys.argv) > 3 else 0, 0, None, 0, None, 0))
SLOTS = 32


def timed(name, flop, run):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        run()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / iters * 1e3
    print("%-40s %8.1f us  %6.1f TF/s  %.2f of peak" % (name, us, flop / us / 1e6, flop / us / 1e6 / 157.3), flush=True)


for P, K, N in [(400000, 128, 256), (524288, 64, 128), (131072, 128, 256), (131072, 128, 128)]:
    x = torch.randn(P, K, device=dev)
    w = torch.randn(N, K, device=dev) * 0.1
    aff = torch.cat([torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1])
    y = torch.empty(P, N, device=dev)
    stats = torch.zeros(SLOTS * 2 * N, dtype=torch.float64, device=dev)

    def fwd():
        rc = so.gb_gemm_fwd(x.data_ptr(), w.data_ptr(), aff.data_ptr(), y.data_ptr(), stats.data_ptr(), SLOTS, P, K, N, None, OPTS, st)
        assert rc == 0, rc
    timed("fwd+affine+stats %d x %d -> %d" % (P, K, N), 2.0 * P * K * N, fwd)

for P, K, N in [(400000, 128, 256), (524288, 64, 128), (131072, 128, 256)]:
    # dX (P,K) = dY (P,N) W (N,K) with the BatchNorm-backward sums of the K-wide layer in front
    dy = torch.randn(P, N, device=dev)
    w = torch.randn(N, K, device=dev) * 0.1
    dx = torch.empty(P, K, device=dev)
    yp = torch.randn(P, K, device=dev)
    ab = torch.cat([torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1, torch.zeros(K, device=dev), torch.ones(K, device=dev)])
    ds = torch.zeros(SLOTS * 2 * K, dtype=torch.float64, device=dev)

    def dgrad():
        rc = so.gb_gemm_dgrad(dy.data_ptr(), w.data_ptr(), dx.data_ptr(), yp.data_ptr(), ab.data_ptr(), ds.data_ptr(), SLOTS, P, K, N,
                              None, None, None, OPTS, st)
        assert rc == 0, rc
    timed("dgrad+BN-backward sums %d x %d -> %d" % (P, N, K), 2.0 * P * K * N, dgrad)

# the crop stacks' pooled last layer: 4096 seeds, ~98 distinct rows each, 4 nested crops
P, K, N, R, D = 400000, 128, 256, 4096, 4
g = torch.Generator(device="cpu").manual_seed(1)
cnt = torch.full((R,), P // R, dtype=torch.int64)
cnt[: P - int(cnt.sum())] += 1
seed = torch.repeat_interleave(torch.arange(R), cnt)
depth = torch.randint(0, D, (P,), generator=g)           # member of crops depth..D-1 (nested)
bits = ((1 << D) - 1) & ~((1 << depth) - 1)
mult = torch.randint(1, 4, (P,), generator=g)
key = ((seed << 13) | (mult << 4) | bits).to(torch.int32)
key = torch.cat([key, torch.zeros(64, dtype=torch.int32)]).to(dev)
x = torch.randn(P, K, device=dev)
w = torch.randn(N, K, device=dev) * 0.1
aff = torch.cat([torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1])
gamma = torch.randn(N, device=dev)
pairs = torch.empty(((P + 31) // 32 + R) * D * N, device=dev)
y = torch.empty(P, N, device=dev)
stats = torch.zeros(SLOTS * 2 * N, dtype=torch.float64, device=dev)


def pool(keep_y):
    def run():
        rc = so.gb_gemm_fwd_pool(x.data_ptr(), w.data_ptr(), aff.data_ptr(), key.data_ptr(), gamma.data_ptr(), pairs.data_ptr(),
                                 pairs.numel(), R, y.data_ptr() if keep_y else None, stats.data_ptr(), SLOTS, P, K, N, D, None, OPTS, st)
        assert rc == 0, rc
    return run


timed("fwd_pool (training: Y kept) %d x %d -> %d" % (P, K, N), 2.0 * P * K * N, pool(True))
timed("fwd_pool (inference: no Y)  %d x %d -> %d" % (P, K, N), 2.0 * P * K * N, pool(False))
