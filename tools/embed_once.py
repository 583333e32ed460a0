import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_embed
r = bench_embed.run(torch.device("cuda:0"), 0, 1, steps=5, warmup=2, with_cpu_baseline=False)
print(r["value"], r["roofline"]["achieved"])
