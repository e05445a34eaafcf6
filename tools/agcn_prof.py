import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _bootstrap; pkg = _bootstrap.load(); import bench
dev='cuda:0'; A = pkg.kinetics_graph().A; shape=(3,300,18,2)
x = torch.rand((64,)+shape, device=dev)
net = pkg.AGcn(A, shape, 400).eval(); bench.randomise_(net, 0); net = net.to(dev)
for _ in range(3): net(x)
torch.cuda.synchronize()
