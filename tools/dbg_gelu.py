import sys, math, numpy as np, torch
sys.path.insert(0, '.')
from simxns_amd import _lib as L
from oracle import bert as ob
dev = torch.device('cuda:0')
M, N, K = 32768, 768, 128
rs = np.random.RandomState(0)
A = (rs.randn(M, K) * 0.5).astype(np.float32); B = (rs.randn(N, K) * 0.5).astype(np.float32); bias = rs.randn(N).astype(np.float32)
bf = lambda a: torch.from_numpy(a).to(dev).to(torch.bfloat16)
dA, dB, dbias = bf(A), bf(B), torch.from_numpy(bias).to(dev)
for trial in range(3):
    C = torch.full((M, N), float('nan'), device=dev, dtype=torch.bfloat16); C2 = torch.full((M, N), float('nan'), device=dev, dtype=torch.bfloat16)
    L.call("simx_gemm_nt", L.stream_ptr(), 1, M, N, K, L.ptr(dA), K, L.ptr(dB), K, L.ptr(C), N, L.ptr(dbias), None, N, 1, None, N, L.ptr(C2), N)
    torch.cuda.synchronize()
    u = C.float().cpu().numpy().astype(np.float64); g = C2.float().cpu().numpy().astype(np.float64)
    ref = ob.gelu(u)
    bad = np.abs(g - ref) > 2e-2 + 2e-2 * np.abs(ref)
    r, c = np.nonzero(bad)
    print("trial", trial, "bad", bad.sum(), "nan", np.isnan(g).sum())
    if bad.sum():
        print(" tile rows:", np.unique(r // 256)[:40], " n tiles cols:", np.unique(c // 256))
        print(" row%256 //16 hist:", np.bincount((r % 256) // 16, minlength=16))
        print(" col%256 //16 hist:", np.bincount((c % 256) // 16, minlength=16))
        print(" row%16 hist:", np.bincount(r % 16, minlength=16))
        print(" col%16 hist:", np.bincount(c % 16, minlength=16))
        i0 = 0
        print(" sample", r[i0], c[i0], g[r[i0], c[i0]], ref[r[i0], c[i0]], u[r[i0], c[i0]])
        # is the bad value the gelu of another element? look for match in same 16x... 
        rr, cc = r[i0], c[i0]
        blk = ref[(rr // 16) * 16:(rr // 16) * 16 + 16, (cc // 64) * 64:(cc // 64) * 64 + 64]
        m = np.argwhere(np.abs(blk - g[rr, cc]) < 1e-2 + 1e-2 * abs(g[rr, cc]))
        print(" matches in same chunk (row, col offsets):", m[:8].tolist(), "own", rr % 16, cc % 64)
        blk2 = ref[((rr // 16) - 1) * 16:((rr // 16)) * 16, (cc // 64) * 64:(cc // 64) * 64 + 64] if rr >= 16 else None
        if blk2 is not None:
            m = np.argwhere(np.abs(blk2 - g[rr, cc]) < 1e-2 + 1e-2 * abs(g[rr, cc])); print(" matches in previous chunk:", m[:8].tolist())
