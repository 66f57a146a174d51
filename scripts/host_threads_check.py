"""Does torch's CPU result (the reference platform) depend on the thread count on this host?
python3 scripts/host_threads_check.py   (CPU only)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "transductive-clip_amd")); sys.path.insert(0, os.path.join(ROOT, "transductive-clip_amd", "drop_in"))
import torch
from oracle import c_oracle, ref_torch
from tclip_amd import synth
K, N, iters = 101, 5, 6
x_q, _ = synth.make_query_tasks(N, K, seed=3052, k_eff=4)
c = c_oracle.run_soft_kmeans(x_q.numpy(), iters=iters, temperature=30)
print("cpu capability", torch.backends.cpu.get_cpu_capability(), "default threads", torch.get_num_threads())
for th in (1, 2, 8, 16, 64, torch.get_num_threads()):
    torch.set_num_threads(th)
    t = ref_torch.run_soft_kmeans(x_q, n_class=K, iters=iters, temperature=30)
    print("threads", th, "entries of u differing from the C++ oracle:", int((c["u"] != t["u"].numpy()).sum()))
