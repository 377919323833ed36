import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import stress_render
for seed in range(51000, 51004):
    try:
        w = stress_render.run(60, seed, verbose=False)
        print("seed", seed, "ok", {k: f"{v:.2e}" for k, v in w.items()}, flush=True)
    except AssertionError as e:
        print("seed", seed, "FAIL", str(e)[:200], flush=True)
