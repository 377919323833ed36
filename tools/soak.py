"""Long randomised soak on the GPU box: the two stress sweeps with fresh seeds, many cases.  usage: python tools/soak.py [cases] [first seed]"""
import sys, os, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import stress_parity, stress_render
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 100
t0 = time.time()
for seed in range(s0, s0 + 4):
    w = stress_parity.run(n, seed, verbose=False)
    print("parity seed", seed, {k: f"{v:.2e}" for k, v in w.items()}, flush=True)
    w = stress_render.run(max(10, n // 2), seed, verbose=False)
    print("render seed", seed, {k: f"{v:.2e}" for k, v in w.items()}, flush=True)
for seed in range(s0, s0 + 4):
    print("dense seed", seed, "long-path cases:", stress_parity.run_dense(max(4, n // 10), seed, verbose=False), flush=True)
print("soak ok in", round(time.time() - t0), "s")
