#!/usr/bin/env python3
"""Per-layer stage timing (HIP events) for the ST-GCN clip path: TFLOP/s of every gcn_stage / tcn_stage launch.
usage: python tools/layer_bench.py [--batch 256] [--iters 5]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import _bootstrap  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--layers", default="1,2,5,6,8,9")
    args = ap.parse_args()
    pkg = _bootstrap.load()
    from continual_skeletons_amd.models import layer_table
    dev = "cuda:0"
    A = pkg.ntu_graph().A
    nm, V = args.batch * 2, 25
    t = 300
    want = {int(x) for x in args.layers.split(",")}
    tot_ms = tot_fl = 0.0
    for i, (ci, co, s, res) in enumerate(layer_table(3)):
        t_out = (t - 1) // s + 1
        if (i + 1) in want:
            blk = pkg.SpatioTemporalBlock(ci, co, A, stride=s, residual=res).eval().to(dev)
            x = torch.rand(nm, ci, t, V, device=dev)
            ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
            g_ms = t_ms = 0.0
            for it in range(args.iters + 1):
                e0, e1, e2 = ev(), ev(), ev()
                e0.record()
                y = blk.gcn(x)
                e1.record()
                ops = blk._packed_ops(x.device)
                mode = 0 if not res else (1 if (ci == co and s == 1) else 2)
                out = pkg.blocks.tcn_stage(y, ops["w"], ops["bias"], ops["c_out"], ops["k"], s, 4, relu=True,
                                           res_mode=mode, x_res=x if mode else None, w_res=ops["w_res"], res_off=0)
                e2.record()
                torch.cuda.synchronize()
                if it:
                    g_ms += e0.elapsed_time(e1)
                    t_ms += e1.elapsed_time(e2)
            g_ms /= args.iters
            t_ms /= args.iters
            r = 4 if ci != co else 3
            g_fl = 2.0 * r * ci * co * t * V * nm
            t_fl = 2.0 * (9 * co * co + (ci * co if mode == 2 else 0)) * t_out * V * nm
            print(f"L{i+1:<2} {ci:>3}->{co:<3} s{s}  gcn {g_ms:7.3f} ms {g_fl/g_ms/1e9:6.1f} TF | tcn {t_ms:7.3f} ms {t_fl/t_ms/1e9:6.1f} TF", flush=True)
            mult = {2: 3, 6: 2, 9: 2}.get(i + 1, 1)     # layers 2-4, 6-7, 9-10 share a shape
            tot_ms += mult * (g_ms + t_ms)
            tot_fl += mult * (g_fl + t_fl)
            del x, y, out, blk
            torch.cuda.empty_cache()
        t = t_out
    print(f"10-block estimate: {tot_ms:.2f} ms / batch {args.batch} -> {args.batch/tot_ms*1e3:.0f} clips/s, {tot_fl/tot_ms/1e9:.1f} TF overall")


if __name__ == "__main__":
    main()
