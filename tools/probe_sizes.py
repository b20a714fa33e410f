import sys, time, torch
sys.path.insert(0, '.')
import bench
dev = torch.device('cuda:0')
from sde_sampler_lrds_amd import _lib as L
def log(*a):
    print(*a, flush=True)
for (B, N) in [(2048, 16), (65536, 4), (65536, 16), (65536, 64), (65536, 256)]:
    t0 = time.perf_counter()
    loss, ts, x0, args, parts, fl = bench.build_rds_gmm(dev, B, N)
    ev = L.HipEvents(); loss.timing_events = ev
    log(f"B={B} N={N} build {time.perf_counter()-t0:.2f}s")
    for rep in range(2):
        t0 = time.perf_counter()
        x, rnd, _ = loss.simulate(ts, x0, *args)
        torch.cuda.synchronize()
        log(f"   simulate wall {time.perf_counter()-t0:.3f}s  kernel {ev.elapsed_ms():.2f} ms  -> {B*N/(ev.elapsed_ms()*1e-3):.3e} p-steps/s  rnd mean {rnd.mean().item():.4f}")
