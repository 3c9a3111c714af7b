# timing-only experiment driver (SLOD_DIAG phase skipping); results are wrong when mask != 0
import os, sys, subprocess, json
import sys as _s
masks = [int(x) for x in _s.argv[1:]] or [0, 1, 2, 4, 8, 16, 32, 63, 59]
for mask in masks:
    env = dict(os.environ, SLOD_DIAG=str(mask))
    out = subprocess.run([sys.executable, "bench.py", "--steps", "5", "--warmup", "1", "--no-cpu-baseline"], env=env, capture_output=True, text=True)
    try:
        j = json.loads(out.stdout.strip().split("\n")[-1])
        print(mask, j["roofline"]["kernel_ms"], flush=True)
    except Exception as e:
        print(mask, "ERR", out.stderr[-500:], flush=True)
