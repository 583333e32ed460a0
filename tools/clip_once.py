import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_embed
print(bench_embed.bench_clip_images(torch.device("cuda:0")))
