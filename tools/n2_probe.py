import sys, json, torch
sys.path.insert(0, '.')
import bench
dev = torch.device('cuda:0')
for a in bench.fused_lines(dev, batches=()):
    print(json.dumps(a))
