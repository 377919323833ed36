"""cProfile of a demo loop on the GPU box (host side): where an iteration's Python time goes.
usage: python tools/demo_host_profile.py ReasonOcclusion|EfficientCuboidViaOptimization [iters]"""
import cProfile, importlib.util, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name = sys.argv[1]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
spec = importlib.util.spec_from_file_location("demo_" + name, os.path.join(ROOT, "demo", name + ".py"))
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)
mod.run(iters=20, log=lambda s: None)      # warm-up (library load, first allocations)
pr = cProfile.Profile()
pr.enable()
out = mod.run(iters=iters)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
