"""What does the conv-residual K phase of the stride-2 TCN stages cost?  tcn_stage with and without the 1x1 strided
residual conv (layers 5 and 8 of the stack, batch 256)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, _bootstrap, bench
pkg = _bootstrap.load()
dev = "cuda:0"; A = pkg.ntu_graph().A
def timed(fn, it=6):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
for (ci, co, t) in [(64, 128, 300), (128, 256, 150)]:
    blk = pkg.SpatioTemporalBlock(ci, co, A, stride=2).eval(); bench.randomise_(blk, 0); blk = blk.to(dev)
    x = torch.rand(512, ci, t, 25, device=dev); y = blk.gcn(x); ops = blk._packed_ops(x.device)
    a = timed(lambda: pkg.blocks.tcn_stage(y, ops["w"], ops["bias"], co, 9, 2, 4, relu=True, res_mode=2, x_res=x, w_res=ops["w_res"]))
    b = timed(lambda: pkg.blocks.tcn_stage(y, ops["w"], ops["bias"], co, 9, 2, 4, relu=True, res_mode=0))
    fl_res = 2.0 * ci * co * (t // 2) * 25 * 512
    print(f"{ci}->{co} s2: with conv residual {a:.3f} ms, without {b:.3f} ms: phase 2 costs {a-b:.3f} ms for {fl_res/1e9:.1f} GFLOP = {fl_res/(a-b)/1e9:.1f} TFLOP/s")
