"""Per-Jacobian time of the exact mode (structured and dense kernels) next to the FD sweep."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from opengoddard_amd import _native, problems            # noqa: E402
from opengoddard_amd.engine import HipEngine             # noqa: E402
import torch                                             # noqa: E402

for name in sys.argv[1:] or ["polar_tsto"]:
    for layout in ("structured", "dense"):
        os.environ.pop("OGPSX_SWEEP", None)
        if layout == "dense":
            os.environ["OGPSX_SWEEP"] = "dense"
        prob, obj = problems.build(name)
        eng = HipEngine(prob, obj)
        n, m = eng.n, eng.m
        x = torch.tensor(np.asarray(prob.p, dtype=np.float64), device="cuda")
        h = torch.full((n,), 1e-8, dtype=torch.float64, device="cuda")
        jt = torch.empty((n, m), dtype=torch.float64, device="cuda")
        f0 = torch.empty((m,), dtype=torch.float64, device="cuda")
        out = {}
        for kind in ("exact", "fd"):
            def run():
                if kind == "exact":
                    eng.exact_dev(x.data_ptr(), 0, n, jt.data_ptr(), f0.data_ptr())
                else:
                    eng.sweep_dev(x.data_ptr(), h.data_ptr(), 0, n, jt.data_ptr(), f0.data_ptr())
            for _ in range(5):
                run()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(200):
                run()
            torch.cuda.synchronize()
            out[kind] = (time.perf_counter() - t) / 200 * 1e6
        print("%-22s %-10s n=%d m=%d exact %.1f us  fd %.1f us" % (name, layout, n, m, out["exact"], out["fd"]),
              flush=True)
        eng.close()
