import ctypes, numpy as np, torch, sys
sys.path.insert(0, '/root/repo')
from sky_embeddings_amd import ops
from oracle import similarity_oracle as so
L = so._lib()
f32p = ctypes.POINTER(ctypes.c_float)
L.skyemb_oracle_dot_variants.argtypes = [f32p, f32p, ctypes.c_int64, f32p]
L.skyemb_oracle_wnorms.argtypes = [f32p, f32p, ctypes.c_int64, ctypes.c_int64, f32p]
rng = np.random.default_rng(0)
for D in (4, 8, 16, 768):
    N = 512
    q = rng.standard_normal((1, D), dtype=np.float32); x = rng.standard_normal((N, D), dtype=np.float32)
    qd, xd = torch.from_numpy(q).cuda(), torch.from_numpy(x).cuda()
    one_q, one_x = torch.ones(1, device='cuda'), torch.ones(N, device='cuda')
    sc = torch.empty(1, N, device='cuda')
    ops.cosine_scores(qd, one_q, xd, one_x, 0.0, sc)
    got = sc.cpu().numpy()[0]
    var = np.zeros((N, 5), np.float32)
    for n in range(N):
        L.skyemb_oracle_dot_variants(q[0].ctypes.data_as(f32p), x[n].ctypes.data_as(f32p), D, var[n].ctypes.data_as(f32p))
    print('D', D, 'match fraction per variant', [(got == var[:, v]).mean() for v in range(5)])
    # norms
    w = (rng.random(D, dtype=np.float32) + 0.1)
    xn = torch.empty(N, device='cuda'); ops.weighted_norms(xd, torch.from_numpy(w).cuda(), xn)
    refn = np.zeros(N, np.float32); L.skyemb_oracle_wnorms(x.ctypes.data_as(f32p), w.ctypes.data_as(f32p), N, D, refn.ctypes.data_as(f32p))
    print('   norms match', (xn.cpu().numpy() == refn).mean())
