"""The headline embed step ONLY (bge-base shape, 256 chunks x 256 tokens, forward + append): the rocprofv3 kernel
statistics of this command are the per-layer table of DESIGN.md section 7 (other shapes: tools/embed_other_once.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_embed
r = bench_embed.run(torch.device("cuda:0"), 0, 1, steps=5, warmup=2, with_cpu_baseline=False, extras=False)
print(r["value"], r["roofline"]["achieved"])
