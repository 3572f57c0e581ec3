"""Where an iteration of fps_rows_kernel goes: in-kernel cycle stamps per phase and wave (diagnostic build:
tools/fps_stamps.sh -> tools/bin/libgraspbal_stamps.so).  Phases: 0 box test, 1 row updates, 2 wave arg-max + publish,
3 barrier wait, 4 cross-wave pick (+ output bookkeeping in wave 0)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graspbalance_amd import _lib
from graspbalance_amd.scene import make_batch
B, N, m = int(os.environ.get('B', 4)), int(os.environ.get('N', 20000)), int(os.environ.get('M', 2048))
L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "libgraspbal_stamps.so"))
P = ctypes.c_void_p
L.gb_fps_pruned.argtypes = [P, P, P, P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint, P, P]
L.gb_fps_row_order.argtypes = [P, P, ctypes.c_int, ctypes.c_int, P]
L.gb_fps_cell_order.argtypes = [P, P, ctypes.c_int, ctypes.c_int, P]
L.gb_debug_fps_stamps.argtypes = [P]
xyz = torch.from_numpy(make_batch(range(B), N)).cuda()
idx = torch.zeros(B, m, dtype=torch.int32, device="cuda")
stamps = torch.zeros(B, 16, 20, dtype=torch.int64, device="cuda")
L.gb_debug_fps_stamps(stamps.data_ptr())
flags = _lib.FPS_SKIP_NEAR_ORIGIN | _lib.FPS_TIE_TREE512
NAMES = {0: "box", 1: "rows:tail(record writes..loop)", 2: "publish(ds_write)", 3: "barrier", 4: "after-pick bookkeeping", 6: "row:read+dist+min",
         7: "row:dpp max+readlane", 8: "row:ballot..ff1", 9: "row:readlanes+writelanes", 10: "argmax:dpp max", 11: "argmax:ballot..readlanes",
         12: "pick:lds read+dpp+readfirstlane", 13: "pick:ballot..readlanes", 14: "(stamp cost)"}
for oname, order in (("rows", L.gb_fps_row_order),):
    perm = torch.empty(B, N, dtype=torch.int32, device="cuda")
    order(xyz.data_ptr(), perm.data_ptr(), B, N, None)
    for lname in os.environ.get("LAYOUTS", "w12").split(","):
        W = int(lname[1:])
        for _ in range(3):
            stamps.zero_()
            rc = L.gb_fps_pruned(xyz.data_ptr(), perm.data_ptr(), None, idx.data_ptr(), B, N, m, flags | _lib.FPS_LAYOUT[lname], None, None)
            assert rc == 0
            torch.cuda.synchronize()
        s = stamps.cpu().double()[:, :W]
        it = s[..., 19].clamp(min=1) - 1
        rows = s[..., 16] / it
        print("order %s layout %s: loop %.0f cycles/iteration (stamped), clock %.0f MHz, rows per wave and iteration %.2f"
              % (oname, lname, float((s[..., 17] / it).mean()), float((s[..., 17] / s[..., 18] * 100).mean()), float(rows.mean())))
        cost = float((s[..., 14] / it).mean())
        for i in sorted(NAMES):
            per = s[..., i] / it
            if i in (6, 7, 8, 9):
                per = s[..., i] / s[..., 16].clamp(min=1)   # per updated row
                print("   %-36s %6.0f cycles per updated row (minus stamp %4.0f)" % (NAMES[i], float(per.mean()), float(per.mean()) - cost))
            else:
                print("   %-36s %6.0f cycles per iteration   (minus stamp %4.0f)" % (NAMES[i], float(per.mean()), float(per.mean()) - cost))
