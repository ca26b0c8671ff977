"""bench.py's CPU sample (the oracle, fp32 eager) at several intra-op thread counts on the GPU box's host: wall time of the sample and clips/s."""
import importlib.util, os, sys, time, types
import torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py"]
spec = importlib.util.spec_from_file_location("bench", os.path.join(root, "bench.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
args = types.SimpleNamespace(frames=8, new_tokens=256)
for n in [int(x) for x in (sys.argv[1:] or ["128", "64", "32", "16"])] if False else (128, 64, 32, 16):
    torch.set_num_threads(n)
    t0 = time.perf_counter(); r = b.cpu_baseline(args)
    print(f"threads {n:3d}: sample {time.perf_counter() - t0:5.1f} s, {r['value']:.5f} clips/s range {r.get('value_range')} | {r['sample'][r['sample'].index('median s'):][:190]}", flush=True)
