# Timing-only experiment driver: SLOD_DIAG=<mask> makes the kernels skip phases (results are
# WRONG when mask != 0; the status check is disabled).  Usage: python tools/phase_timing.py 0 16 20
# k_solve_ws bits: 2 RHS build, 4 sweep (1 pivot only), 8 GEMM, 16 backward pass, 32 band loads,
# 16384 next_S, 32768 store_V.  k_select bits: 64 M, 128 D, 256 BD fill, 512 QR+SVD, 1024 phi,
# 2048 psi, 4096 SVD fallback, 8192 cap sweeps at 3.
import os, sys, subprocess, json
import sys as _s
masks = [int(x) for x in _s.argv[1:]] or [0, 1, 2, 4, 8, 16, 32, 63, 59]
for mask in masks:
    env = dict(os.environ, SLOD_DIAG=str(mask))
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), "--steps", "5", "--warmup", "1", "--no-cpu-baseline", "--no-pipeline"], env=env, capture_output=True, text=True)
    try:
        j = json.loads(out.stdout.strip().split("\n")[-1])
        print(mask, j["roofline"]["kernel_ms"], flush=True)
    except Exception as e:
        print(mask, "ERR", out.stderr[-500:], flush=True)
