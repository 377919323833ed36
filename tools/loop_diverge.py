"""Does the replayed ShapeFitting loop follow the eager one?  (debug aid)  usage: python tools/loop_diverge.py [sync|nosync]"""
import importlib.util, os, sys, numpy as np, torch
sys.path.insert(0, ".")
spec = importlib.util.spec_from_file_location("sf", os.path.join("demo", "ShapeFitting.py"))
sf = importlib.util.module_from_spec(spec); spec.loader.exec_module(sf)
mode = sys.argv[1] if len(sys.argv) > 1 else "nosync"
orig = sf.BatchedIteration.__call__
rec = []
def dbg(self, views, w):
    out = orig(self, views, w)
    if mode == "sync":
        torch.cuda.synchronize()
    if len(rec) < 4:
        ps = self._params()
        rec.append((out.clone(), ps[0].detach().abs().max().clone(), ps[0].grad.abs().max().clone(), ps[1].grad.abs().max().clone(),
                    self.optimizer.state[ps[0]]["momentum_buffer"].abs().max().clone()))
    return out
sf.BatchedIteration.__call__ = dbg
h = sf.fit(quiet=True, iters=6, rgb_on=40, graph=True)
for i, r in enumerate(rec):
    print("call", i, "losses", r[0].tolist(), "|verts|", float(r[1]), "|g verts|", float(r[2]), "|g col|", float(r[3]), "|mom|", float(r[4]))
print(mode, np.asarray(h["silhouette"]).round(6).tolist(), flush=True)
