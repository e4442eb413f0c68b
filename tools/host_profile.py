#!/usr/bin/env python3
"""Where the host time of an eagerly launched config-3 step goes (cProfile over 50 steps, top entries by own time)."""
import cProfile
import os
import pstats
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]]
import runpy
ns = runpy.run_path(os.path.join(ROOT, 'tools', 'bench_net.py'))
import torch
step = ns['step']
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(28)
st.sort_stats('cumtime').print_stats(45)
