"""Run the eleven differential fuzzers in ONE process (one torch import, one library load): `python scripts/fuzz_all.py [scale]` - scale 1 = the short
fixed-seed runs of the GPU suite (tests/test_fuzz_gpu.py), larger = proportionally more cases.  Prints one RESULT line per fuzzer; exit code 1 if any failed."""
import io, os, runpy, sys, time, contextlib
HERE = os.path.dirname(os.path.abspath(__file__))
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
RUNS = [("fuzz_gemm.py", 4000, 11), ("fuzz_attn.py", 1500, 12), ("fuzz_decoder.py", 8, 13), ("fuzz_multimodal.py", 6, 14), ("fuzz_rope_epilogue.py", 400, 15),
        ("fuzz_frontend.py", 60, 16), ("fuzz_ops.py", 2000, 17), ("fuzz_seg.py", 1200, 18), ("fuzz_engine_state.py", 120, 19), ("fuzz_model_state.py", 30, 20), ("fuzz_vqgan.py", 24, 21)]
failed = 0
for script, cases, seed in RUNS:
    argv = sys.argv
    sys.argv = [script, str(max(1, int(cases * scale))), str(seed)]
    buf = io.StringIO()
    t0 = time.time()
    rc = 0
    try:
        with contextlib.redirect_stdout(buf):
            runpy.run_path(os.path.join(HERE, script), run_name="__main__")
    except SystemExit as e:
        rc = int(e.code or 0)
    except Exception as e:      # noqa: BLE001
        rc = 2
        buf.write(f"\n{type(e).__name__}: {e}\n")
    finally:
        sys.argv = argv
    out = buf.getvalue().strip().splitlines()
    summary = next((l for l in reversed(out) if "failures" in l), out[-1] if out else "")
    print(f"RESULT {script} rc={rc} ({time.time() - t0:.1f} s): {summary}", flush=True)
    if rc:
        failed += 1
        print("\n".join(out[-25:]), flush=True)
sys.exit(1 if failed else 0)
