"""Kernel-only durations of the few-row GEMM calls under rocprofv3 --kernel-trace: ring (csrc/gemm_ring.hip) vs the
register-staged tiles (GB_GEMM_NO_RING), per shape and op.  A one-element fill marks the boundary between groups.
   rocprofv3 --kernel-trace --output-format csv -d DIR -o b -- python3 tools/ring_prof.py run
   python3 tools/ring_prof.py parse DIR/.../b_kernel_trace.csv"""
import ctypes, csv, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = [(4096, 1024, 256), (4096, 256, 1024), (4096, 256, 256), (8192, 512, 128), (8192, 128, 512), (8192, 128, 128),
          (2048, 1024, 256), (2048, 256, 1024), (2048, 256, 256), (1024, 1024, 256), (1024, 256, 1024), (1024, 256, 256),
          (16384, 1024, 256), (16384, 256, 128), (16384, 128, 128), (4096, 256, 304)]
if os.environ.get("RING_PROF_SHAPES") == "tall":   # the 16 384 - 32 768-row products: ring vs LDS tiles vs row streaming
    SHAPES = [(16384, 128, 128), (16384, 256, 128), (16384, 128, 256), (32768, 128, 128), (32768, 128, 256), (16384, 1024, 256),
              (65536, 128, 128), (65536, 128, 256)]
OPS = ("fwd+aff+stats", "fwd", "dgrad+bn", "dgrad", "wgrad+aff")
VARIANTS = ("ring", "tiles", "rows") if os.environ.get("RING_PROF_SHAPES") == "tall" else ("ring", "tiles")
REPS = 10
if sys.argv[1] == "run":
    import torch
    from graspbalance_amd import _lib as L
    lib = L.lib()
    dev = "cuda:0"
    ws = torch.empty(L.GEMM_SCRATCH_BYTES, dtype=torch.uint8, device=dev)
    mk_x = torch.zeros(3, device=dev); mk_m = torch.zeros(12, dtype=torch.float64, device=dev)
    mk_c = torch.zeros(1, dtype=torch.int32, device=dev); mk_o = torch.zeros(1, dtype=torch.int64, device=dev)
    mk_t = torch.zeros(1, dtype=torch.int64, device=dev)
    begin = lambda: lib.gb_moments3(L.ptr(mk_x), None, 1, L.ptr(mk_m), None, None)          # moments3_kernel
    end = lambda: lib.gb_cyl_scan(L.ptr(mk_c), 1, 1, L.ptr(mk_o), L.ptr(mk_t), None)       # cyl_scan_kernel
    for P, K, N in SHAPES:
        X = torch.randn(P, K, device=dev); W = torch.randn(N, K, device=dev); Y = torch.empty(P, N, device=dev)
        dY = torch.randn(P, N, device=dev); dX = torch.empty(P, K, device=dev); dW = torch.zeros(N, K, device=dev)
        aff = torch.randn(2 * K, device=dev); st = torch.zeros(2 * N, dtype=torch.float64, device=dev)
        ab = torch.randn(4 * K, device=dev); dst = torch.zeros(2 * K, dtype=torch.float64, device=dev)
        rows_dev = torch.tensor([P], dtype=torch.int64, device=dev)
        for var in VARIANTS:
            flags = L.GEMM_NO_RING if var == "tiles" else 0
            # ("rows": a device-side row count forces the row-streaming kernel for forward / dgrad)
            o = ctypes.pointer(L.GemmOpts(L.PREC_F32, 0, ws.data_ptr(), ws.numel(), rows_dev.data_ptr() if var == "rows" else None, flags))
            calls = (lambda: lib.gb_gemm_fwd(L.ptr(X), L.ptr(W), L.ptr(aff), L.ptr(Y), L.ptr(st), 1, P, K, N, None, o, None),
                     lambda: lib.gb_gemm_fwd(L.ptr(X), L.ptr(W), None, L.ptr(Y), None, 1, P, K, N, None, o, None),
                     lambda: lib.gb_gemm_dgrad(L.ptr(dY), L.ptr(W), L.ptr(dX), L.ptr(X), L.ptr(ab), L.ptr(dst), 1, P, K, N, None, None, None, o, None),
                     lambda: lib.gb_gemm_dgrad(L.ptr(dY), L.ptr(W), L.ptr(dX), None, None, None, 0, P, K, N, None, None, None, o, None),
                     lambda: lib.gb_gemm_wgrad(L.ptr(dY), L.ptr(X), L.ptr(aff), L.ptr(dW), P, K, N, o, None))
            for fn in calls:
                ok = fn() == 0      # (a variant that cannot run a shape - e.g. "rows" with a weight matrix beyond the LDS -
                fn()                #  leaves an empty group: 0.0 in the table)
                torch.cuda.synchronize()
                begin()
                for _ in range(REPS if ok else 0):
                    assert fn() == 0
                torch.cuda.synchronize()
                end()
    torch.cuda.synchronize()
else:
    rows = list(csv.DictReader(open(sys.argv[2])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    groups, cur, inside = [], 0.0, False
    names = set()
    for r in rows:
        name = r["Kernel_Name"]
        if "moments3_kernel" in name:
            cur, inside = 0.0, True
        elif "cyl_scan_kernel" in name:
            if inside:
                groups.append(cur / REPS / 1e3)
            inside = False
        elif inside and "gb::" in name:
            cur += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    want = len(SHAPES) * len(VARIANTS) * len(OPS)
    print("groups", len(groups), "expected", want)
    it = iter(groups)
    print("%-20s" % "shape" + "".join("%26s" % o for o in OPS) + "   (us: " + " / ".join(VARIANTS) + ")")
    tot = [[0.0] * len(VARIANTS) for _ in OPS]
    for shp in SHAPES:
        v = [[next(it) for _ in OPS] for _ in VARIANTS]
        for i in range(len(OPS)):
            for j in range(len(VARIANTS)):
                tot[i][j] += v[j][i]
        print("%-20s" % str(shp) + "".join("   " + " /".join("%7.1f" % v[j][i] for j in range(len(VARIANTS))) for i in range(len(OPS))))
    print("%-20s" % "sum" + "".join("   " + " /".join("%7.1f" % t for t in tot[i]) for i in range(len(OPS))))
