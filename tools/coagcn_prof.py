"""rocprofv3 target: CoAGCN online cycles (Kinetics shape, 1024 streams, 4 frames per launch)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _bootstrap; pkg = _bootstrap.load(); import bench
dev = 'cuda:0'; A = pkg.kinetics_graph().A; shape = (3, 300, 18, 2)
streams = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
frames = torch.rand((8, streams, 3, 18, 2), device=dev)
net = (pkg.CoStGcn if len(sys.argv) > 2 else pkg.CoAGcn)(A, shape, 400).eval(); bench.randomise_(net, 0); net = net.to(dev)
for c in range(40):
    net.forward_cycle([frames[(4 * c + f) % 8] for f in range(4)])
torch.cuda.synchronize()
