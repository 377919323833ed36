"""Dense-object soak (the binning's long path, extensions, slice re-test), many seeds: races show up as a rare mismatch.
usage: python tools/soak_dense.py [seeds] [first seed] [cases per seed]"""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import stress_parity
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 20
s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
nc = int(sys.argv[3]) if len(sys.argv) > 3 else 12
t0 = time.time(); tot = 0
for seed in range(s0, s0 + ns):
    tot += stress_parity.run_dense(nc, seed, verbose=False)
print(f"dense soak ok: {ns * nc} cases, {tot} on the long path, {time.time() - t0:.0f} s")
