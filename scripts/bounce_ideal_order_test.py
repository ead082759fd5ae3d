"""Ceiling for any re-ordering of bounce rays: first-bounce rays of C2 sorted by their own ANSWER (hit triangle, then
origin cell for the misses) -- an order no pre-pass could know -- against path order.  usage: ... 1 8"""
import os, sys
import numpy as np, torch
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bounce_sort_test.py")).read().split("octant = ")[0])
d_p = torch.empty(m, dtype=torch.int32, device="cuda"); d_t = torch.empty(m, dtype=torch.float32, device="cuda")
a = torch.from_numpy(o).cuda(); b = torch.from_numpy(d).cuda()
s.trace_closest_device(a.data_ptr(), b.data_ptr(), m, d_p.data_ptr(), d_t.data_ptr())
prim = d_p.cpu().numpy().astype(np.int64)
key = np.where((prim >= 0) & (prim < 81920), prim, 81920 + morton(o, 6))
for name, perm in (("as generated", np.arange(m)), ("by answer", np.argsort(key, kind="stable")),
                   ("by answer, then direction", np.argsort(key * 64 + dir_cell(d, 8), kind="stable"))):
    a = torch.from_numpy(np.ascontiguousarray(o[perm])).cuda(); b = torch.from_numpy(np.ascontiguousarray(d[perm])).cuda()
    for _ in range(3):
        st = s.trace_closest_device(a.data_ptr(), b.data_ptr(), m, d_p.data_ptr(), d_t.data_ptr())
    print("%-26s Mrays/s %.1f" % (name, m / st["ms_extend"] / 1e3), flush=True)
