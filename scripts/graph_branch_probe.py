"""Does a captured hipGraph run independent branches concurrently on this ROCm?  Two chains of small kernels (each far from filling
256 CUs): one stream / two streams eager / two streams captured."""
import torch, time
torch.cuda.init()
n = 1 << 16                                  # 64K elements: 64 workgroups of 1024 -> quarter of the CUs
a = torch.randn(n, device="cuda"); b = torch.randn(n, device="cuda")
def chain(x, k=300):
    for _ in range(k):
        x = torch.sin(x) * 1.0001            # two tiny kernels per iteration
    return x
def two_streams():
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1): r1 = chain(a)
    with torch.cuda.stream(s2): r2 = chain(b)
    cur.wait_stream(s1); cur.wait_stream(s2)
    return r1, r2
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t = time.time()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.time() - t) / reps * 1e3
print("one stream, eager      : %.2f ms" % timeit(lambda: (chain(a), chain(b))))
print("two streams, eager     : %.2f ms" % timeit(two_streams))
g1 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g1): chain(a); chain(b)
print("one stream, graph      : %.2f ms" % timeit(g1.replay))
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2): two_streams()
print("two branches, graph    : %.2f ms" % timeit(g2.replay))
# two SEPARATE graphs replayed on two streams (does hipGraphLaunch on different streams overlap?)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
with torch.cuda.graph(ga, stream=s1): chain(a)
with torch.cuda.graph(gb, stream=s2): chain(b)
def two_graphs():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1): ga.replay()
    with torch.cuda.stream(s2): gb.replay()
    cur.wait_stream(s1); cur.wait_stream(s2)
print("two graphs, two streams: %.2f ms" % timeit(two_graphs))
# bigger kernels: one latency-bound chain (tiny kernels) next to one bandwidth-bound chain (256 MB elementwise passes)
big = torch.randn(1 << 26, device="cuda")
def bw_chain(k=20):
    x = big
    for _ in range(k):
        x = x * 1.0001
    return x
gl, gw = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
with torch.cuda.graph(gl, stream=s1): chain(a, 600)
with torch.cuda.graph(gw, stream=s2): bw_chain()
def lat_only():
    with torch.cuda.stream(s1): gl.replay()
    torch.cuda.current_stream().wait_stream(s1)
def bw_only():
    with torch.cuda.stream(s2): gw.replay()
    torch.cuda.current_stream().wait_stream(s2)
def both():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1): gl.replay()
    with torch.cuda.stream(s2): gw.replay()
    cur.wait_stream(s1); cur.wait_stream(s2)
print("latency-bound graph alone: %.2f ms, bandwidth-bound graph alone: %.2f ms, both on two streams: %.2f ms" % (timeit(lat_only), timeit(bw_only), timeit(both)))
