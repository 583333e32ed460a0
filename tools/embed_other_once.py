"""The other embed shapes (bge S=512, MiniLM S=256, log-normal lengths), one pass each: kernel statistics of the shapes
that are NOT the headline (kept apart from tools/embed_once.py so that neither CSV mixes shapes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_embed
print(bench_embed.bench_other_shapes(torch.device("cuda:0")))
