import os, sys, json, subprocess
for mode in ("0", "1", "2", "3", "4"):
    env = dict(os.environ, VOGE_TRACE_MODE=mode)
    out = subprocess.run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"], env=env, capture_output=True, text=True).stdout
    d = json.loads(out.strip().splitlines()[-1])
    print("mode", mode, "trace_fwd ms", d["stages"]["trace_fwd"]["ms"])
