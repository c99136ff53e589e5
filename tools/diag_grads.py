import os, sys, numpy as np, torch
ROOT=os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_encoder_gpu as T
for name in ("step_tiny.npz", "step_base_cfg1.npz"):
    G = np.load(os.path.join(ROOT, "tests", "golden", name))
    R = T.run_step(G, torch.device("cuda:0"), "fp32")
    rows = []
    if "grad_names" in G.files:
        names = [str(n) for n in G["grad_names"]]; norms = G["grad_norms"]
        for n, ref in zip(names, norms):
            got = np.sqrt((R["grads"][n] ** 2).sum()); rows.append((abs(got-ref)/max(ref,1e-30), ref, abs(got-ref), n))
        print(name, "max norm", norms.max())
    else:
        gm = max(np.abs(G["grad." + k]).max() for k in R["grads"])
        print(name, "global max", gm)
        for k, g in R["grads"].items():
            ref = G["grad." + k]; rows.append((np.abs(g-ref).max()/max(np.abs(ref).max(),1e-30), np.abs(ref).max(), np.abs(g-ref).max(), k))
    rows.sort(reverse=True)
    for r in rows[:8]: print("  rel %.3e scale %.3e abs %.3e %s" % r)
    print("  q err", np.abs(R["q"]-G["q_emb"]).max(), "sim err", np.abs(R["sim"]-G["sim"]).max(), "loss err", abs(R["loss"]-float(G["loss_kl"])))
