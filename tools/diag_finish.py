import ctypes, numpy as np, torch, sys
sys.path.insert(0, '/root/repo')
from sky_embeddings_amd import ops
from oracle import similarity_oracle as so
rng = np.random.default_rng(0)
Q, N, D = 3, 1024, 96
q = rng.standard_normal((Q, D), dtype=np.float32); x = rng.standard_normal((N, D), dtype=np.float32)
w = rng.random(D, dtype=np.float32) + 0.1; w /= w.sum()
qd, xd, wd = torch.from_numpy(q).cuda(), torch.from_numpy(x).cuda(), torch.from_numpy(w).cuda()
tw, qn, xn = torch.empty(Q, D, device='cuda'), torch.empty(Q, device='cuda'), torch.empty(N, device='cuda')
ops.weighted_norms(qd, wd, qn, tw); ops.weighted_norms(xd, wd, xn)
print('tw equal', np.array_equal(tw.cpu().numpy(), w[None] * q))
one_q, one_x = torch.ones(Q, device='cuda'), torch.ones(N, device='cuda')
dot = torch.empty(Q, N, device='cuda'); ops.cosine_scores(tw, one_q, xd, one_x, 0.0, dot)
sc = torch.empty(Q, N, device='cuda'); ops.cosine_scores(tw, qn, xd, xn, 1e-6, sc)
dot, sc, qn_, xn_ = dot.cpu().numpy(), sc.cpu().numpy(), qn.cpu().numpy(), xn.cpu().numpy()
den = (qn_[:, None] * xn_[None, :]).astype(np.float32) + np.float32(1e-6)
ref = dot / den
print('finish equal (numpy sep mul/add/div)', (ref == sc).mean())
den2 = (qn_[:, None].astype(np.float64) * xn_[None, :].astype(np.float64) + 1e-6).astype(np.float32)
print('finish equal (fused mul-add)', ((dot / den2) == sc).mean())
ref_sc = so.cosine_scores_np(q, x, w)
print('oracle equal', (ref_sc == sc).mean(), 'oracle vs numpy finish', (ref_sc == ref).mean())
# oracle pieces
L = so._lib(); f32p = ctypes.POINTER(ctypes.c_float)
L.skyemb_oracle_wnorms.argtypes = [f32p, f32p, ctypes.c_int64, ctypes.c_int64, f32p]
refn = np.zeros(N, np.float32); L.skyemb_oracle_wnorms(x.ctypes.data_as(f32p), w.ctypes.data_as(f32p), N, D, refn.ctypes.data_as(f32p))
print('xn equal', (refn == xn_).mean())
one = np.ones(D, np.float32)
d_or = so.cosine_scores_np(q, x, w)  # placeholder
print('max abs diff oracle-gpu', np.abs(ref_sc - sc).max(), 'eps check', np.float32(1e-6))
