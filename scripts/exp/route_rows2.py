"""The decode-regime router at 512 rows: two rows per block (r05) against one (CRAB_ROUTE_ROWS=1), us per launch over rotating inputs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from crab_amd import ops
BF = torch.bfloat16
def timeit(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M in (448, 512):
    for name, K, nproj in (("o group", 4096, 1), ("down group", 11008, 1), ("q|k|v group", 4096, 3), ("gate|up group", 4096, 2)):
        nl, r = 3, 8
        tcols = (nproj * (nl + r) + 15) // 16 * 16
        xs = [(torch.randn(M, K, device="cuda") * 0.7).to(BF) for _ in range(8)]
        ra = torch.zeros(tcols, K, device="cuda", dtype=BF); ra[: nproj * (nl + r)] = (torch.randn(nproj * (nl + r), K, device="cuda") * 0.02).to(BF)
        ucols = (nproj * nl * r + 7) // 8 * 8
        u = torch.empty(M, ucols, device="cuda", dtype=BF)
        ws = torch.empty(ops.hyperlora_route_workspace(M, K, tcols), device="cuda", dtype=torch.uint8)
        res = {}
        for env in ("1", None, "1", None):
            if env: os.environ["CRAB_ROUTE_ROWS"] = env
            else: os.environ.pop("CRAB_ROUTE_ROWS", None)
            i = [0]
            def fn():
                i[0] = (i[0] + 1) % 8
                ops.hyperlora_route(xs[i[0]], ra, nproj, nl, r, ucols, 2.0, out=u, workspace=ws)
            res.setdefault(env or "2", []).append(timeit(fn))
        os.environ.pop("CRAB_ROUTE_ROWS", None)
        print(f"M={M} {name:14s} K={K:5d} nproj={nproj}: one row per block {min(res['1']):6.1f} us, two rows {min(res['2']):6.1f} us", flush=True)
