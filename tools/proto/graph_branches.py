"""Does a HIP graph with two parallel branches replay faster than the same kernels in one chain?  Chain A: 16 small kernels (one
under-filled 'conv-like' kernel each, ~10-20 us); chain B: 16 tiny element-wise kernels.  Serial capture vs B forked onto a second stream."""
import torch, time
dev = torch.device('cuda')
a = torch.randn(4, 256, 256, device=dev)          # tiny bmm: a few workgroups, latency-bound
b = torch.randn(4, 256, 256, device=dev)
x = torch.randn(1 << 16, device=dev)
def chain_a(n=16):
    y = a
    for _ in range(n): y = torch.bmm(y, b) * 0.04
    return y
def chain_b(n=16):
    z = x
    for _ in range(n): z = z * 1.0001 + 0.5
    return z
def capture(fork):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(); s2 = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): chain_a(); chain_b()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            if fork:
                s2.wait_stream(s)
                with torch.cuda.stream(s2):
                    zb = chain_b()
                ya = chain_a()
                s.wait_stream(s2)
            else:
                ya = chain_a(); zb = chain_b()
    return g
for fork in (False, True, False, True):
    g = capture(fork)
    for _ in range(5): g.replay()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(200): g.replay()
    torch.cuda.synchronize()
    print('fork' if fork else 'serial', f'{(time.perf_counter() - t) / 200 * 1e6:.1f} us per replay')
