import sys, os, math, torch
sys.path.insert(0, os.getcwd())
from diffusion_model_amd.engine import Engine
dev = torch.device("cuda", 0); dt = torch.bfloat16
eng = Engine.get(dev, dt)
c = 320
torch.manual_seed(0)
for m in (128, 768, 2048, 4096):
    eng.reset()
    src = torch.randn(m, 64, device=dev).to(dt)
    t = eng.gemm(src, eng.pack_linear(torch.nn.Linear(64, c).to(dev)), ln_out=True, persistent=True)
    rows = eng.ln_stats(t)
    x = torch.randn(m, c, device=dev).to(dt)
    p1 = eng.pack_linear(torch.nn.Linear(c, 8 * c).to(dev), geglu=True, ln=torch.nn.LayerNorm(c).to(dev))
    p2 = eng.pack_chain(torch.nn.Linear(4 * c, c).to(dev), torch.nn.Conv2d(c, c, 1).to(dev))
    g = eng.gemm(t, p1, ln_in=rows)
    ref = eng.gemm(g, p2, a2=t, residual=x).float()
    outs = []
    for rep in range(3):
        out = torch.zeros(m, c, dtype=dt, device=dev)
        eng.ffn_geglu(t, p1, p2, out, residual=x, ln_in=rows)
        outs.append(out.float())
    d = (outs[0] - ref).abs()
    bad = d > 0.05 * ref.abs().max()
    print("m", m, "maxdiff", float(d.max()), "bad", int(bad.sum()), "deterministic", bool(torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])))
    for a in range(3):
        dd = (outs[a] - ref).abs()
        nz = dd > 1e-6
        r, cc = nz.nonzero(as_tuple=True)
        print("  rep", a, "differs from two-launch in", int(nz.sum()), "max", float(dd.max()),
              "rowgrp:", torch.bincount((r % 128) // 32, minlength=4).tolist() if len(r) else [],
              "col//64:", torch.bincount(cc // 64, minlength=5).tolist() if len(r) else [],
              "(col%16)//8:", torch.bincount((cc % 16) // 8, minlength=2).tolist() if len(r) else [],
              "blocks:", torch.unique(r // 128).tolist()[:12])
    if bad.any():
        r, cc = bad.nonzero(as_tuple=True)
        print("  bad rows (mod 128) hist of row//32:", torch.bincount((r % 128) // 32, minlength=4).tolist(), " blocks:", torch.unique(r // 128).tolist()[:20])
        print("  bad cols hist of col//64:", torch.bincount(cc // 64, minlength=5).tolist(), " (col%16)//8:", torch.bincount((cc % 16) // 8, minlength=2).tolist())
