import cProfile, pstats, sys, os, io
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
from __graft_entry__ import load_package
pkg = load_package(); torch.cuda.set_device(0)
import bench_secondary as bs
pr = cProfile.Profile()
pr.enable()
r = bs.feast_3d(pkg, torch, 80)
pr.disable()
print(r["value"], r["stage_seconds"])
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:9000])
