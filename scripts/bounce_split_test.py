"""Cost of first-bounce rays by where they start (C2): rays leaving the sphere never hit it again (convex) and still
cost 40 % of the batch -- a descent from the root to their own leaf.  usage: python scripts/bounce_split_test.py 1 8"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bounce_sort_test.py")).read().split("def dir_cell")[0].replace('sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))',''))
r = np.linalg.norm(o - c, axis=1)
on_sphere = np.abs(r - 2.5) < 0.01
d_p = torch.empty(m, dtype=torch.int32, device="cuda"); d_t = torch.empty(m, dtype=torch.float32, device="cuda")
for name, sel in (("all", np.ones(m, bool)), ("from the sphere", on_sphere), ("from the walls", ~on_sphere)):
    oo, dd = np.ascontiguousarray(o[sel]), np.ascontiguousarray(d[sel]); k = len(oo)
    a = torch.from_numpy(oo).cuda(); b = torch.from_numpy(dd).cuda()
    for _ in range(3):
        st = s.trace_closest_device(a.data_ptr(), b.data_ptr(), k, d_p.data_ptr(), d_t.data_ptr())
    hits = int((d_p[:k] < 81920).logical_and(d_p[:k] >= 0).sum())
    print("%-16s rays %9d  Mrays/s %.1f  ms %.2f  mesh-hit frac %.3f" % (name, k, k / st["ms_extend"] / 1e3, st["ms_extend"], hits / k), flush=True)
