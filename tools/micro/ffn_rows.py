import sys, os, ctypes as C, torch
sys.path.insert(0, os.getcwd())
from diffusion_model_amd.engine import Engine
dev = torch.device("cuda", 0); dt = torch.bfloat16
eng = Engine.get(dev, dt)
c = 320
def graph_time(fn, reps=10):
    stream = eng.stream
    fn(); torch.cuda.synchronize()
    graph = C.c_void_p()
    assert eng.lib.dm_graph_begin(stream) == 0
    for _ in range(reps): fn()
    assert eng.lib.dm_graph_end(stream, C.byref(graph)) == 0
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); eng.lib.dm_graph_launch(graph, stream); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) / reps * 1e3)
    eng.lib.dm_graph_destroy(graph)
    return sorted(ts)[2]
side = torch.cuda.Stream(device=dev)
with torch.cuda.stream(side):
    for m in (128, 256, 1024, 4096, 32768, 65536, 131072):
        eng.reset()
        src = torch.randn(m, 64, device=dev).to(dt)
        t = eng.gemm(src, eng.pack_linear(torch.nn.Linear(64, c).to(dev)), ln_out=True, persistent=True)
        rows = eng.ln_stats(t)
        x = torch.randn(m, c, device=dev).to(dt)
        p1 = eng.pack_linear(torch.nn.Linear(c, 8 * c).to(dev), geglu=True, ln=torch.nn.LayerNorm(c).to(dev))
        p2 = eng.pack_chain(torch.nn.Linear(4 * c, c).to(dev), torch.nn.Conv2d(c, c, 1).to(dev))
        out = torch.empty(m, c, dtype=dt, device=dev)
        tf = graph_time(lambda: eng.ffn_geglu(t, p1, p2, out, residual=x, ln_in=rows))
        print(f"m={m:7d} blocks={m//128:5d}: {tf:8.1f} us", flush=True)
