import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sde_sampler_lrds_amd.experiments import baseline_configs as cfgs
from oracle import baseline_oracles as bo, sde_oracle as orc
dev = torch.device("cuda:0")
for N in (2, 8, 40):
    loss, ts, x0, args, kw, info = cfgs.build_rds_gmm(dev, 64, N, d=128, K=4, seed=132)
    loss.seed = 13
    loss.split_tiles = False
    a = loss.simulate(ts, x0, *args)
    loss.split_tiles = True
    b = loss.simulate(ts, x0, *args)
    run = bo.runner("rds_gmm", info, ts)
    ox, ornd, sc = run(x0.cpu(), orc.PhiloxNoise(13))
    # fp64 oracle of the same trajectory: which fp32 result is closer to exact arithmetic?
    ea = (a[0].cpu() - ox).abs().amax(1); eb = (b[0].cpu() - ox).abs().amax(1)
    bad = (eb > 1e-5).nonzero().flatten().tolist()
    print(f"N={N}: standard vs oracle max {float(ea.max()):.2e}; split vs oracle max {float(eb.max()):.2e}; rows where split is off: {bad[:10]}; standard's error there: {[f'{float(ea[i]):.1e}' for i in bad[:10]]}")
    print("   rnd: standard vs oracle", float((a[1].cpu() - ornd).abs().max()), " split vs oracle", float((b[1].cpu() - ornd).abs().max()))
