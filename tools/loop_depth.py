"""How the replayed ShapeFitting iteration behaves under different host queue depths (debug aid).
usage: python tools/loop_depth.py"""
import importlib.util, os, sys, time, torch
sys.path.insert(0, ".")
spec = importlib.util.spec_from_file_location("sf", os.path.join("demo", "ShapeFitting.py"))
sf = importlib.util.module_from_spec(spec); spec.loader.exec_module(sf)
for graph in (True, False):
    for iters, settle in ((300, 0), (300, 300), (300, 2000), (3000, 0)):
        h = sf.fit(iters=settle + iters, timed_from=settle, quiet=True, rgb_on=0, graph=graph)
        print(f"graph={graph} timed {iters} after {settle} untimed: {h['sec_per_iter'] * 1e3:.4f} ms per iteration", flush=True)
