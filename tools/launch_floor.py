import torch
dev = torch.device('cuda',0)
x = torch.zeros(4096, dtype=torch.float64, device=dev)
y = torch.zeros(4096*5*2, dtype=torch.float64, device=dev)
for name, fn in (("add_ 4096 f64", lambda: x.add_(1.0)), ("add_ 40960 f64", lambda: y.add_(1.0)), ("sqrt chain", lambda: y.copy_(torch.sqrt(y*y+1)))):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(200): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        s.record(); g.replay(); e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e)/200)
    print("%-16s %.2f us per launch (graph of 200)" % (name, best*1e3))
