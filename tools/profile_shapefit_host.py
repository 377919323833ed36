import cProfile, pstats, sys, os, importlib.util
sys.path.insert(0, '.')
spec = importlib.util.spec_from_file_location("d", "demo/ShapeFitting.py"); d = importlib.util.module_from_spec(spec); spec.loader.exec_module(d)
d.fit(iters=20, quiet=True)
pr = cProfile.Profile(); pr.enable()
h = d.fit(iters=200, quiet=True)
pr.disable()
print("ms/iter", h["sec_per_iter"]*1e3)
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
