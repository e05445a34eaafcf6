import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _bootstrap; pkg = _bootstrap.load()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import bench
dev = 'cuda:0'
net = pkg.StGcn(pkg.ntu_graph().A).eval(); bench.randomise_(net, 0); net = net.to(dev)
x = torch.rand((256, 3, 300, 25, 2), device=dev)
def run(nsplit, iters=6):
    parts = [p.contiguous() for p in x.chunk(nsplit)]
    streams = [torch.cuda.Stream() for _ in range(nsplit)]
    def step():
        cur = torch.cuda.current_stream()
        outs = []
        for p, s in zip(parts, streams):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                outs.append(net(p))
        for s in streams: cur.wait_stream(s)
        return torch.cat(outs)
    for _ in range(2): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): o = step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / iters
    return dt
for n in (1, 2, 4):
    print(n, 'streams:', f'{run(n)*1e3:.2f} ms/step', f'{256/run(n):.0f} clips/s')
