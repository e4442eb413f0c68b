import os, sys, runpy, torch
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, ROOT)
ns = runpy.run_path(os.path.join(ROOT, 'tools', 'bench_net.py'))
step = ns['step']
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], record_shapes=False) as prof:
    for _ in range(20):
        step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=40, max_name_column_width=60))
