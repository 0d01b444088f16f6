"""Frame-by-frame comparison of the HIP path with the oracle on the inverse-depth-bound scene of tests/test_gpu_depth_holes.py: iteration counts,
accepted steps, line-search trial / contraction counts, clamp counts and position distance per frame (GPU box; test infrastructure)."""
import sys
import os; ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,os.path.join(ROOT,"tests")); sys.path.insert(0,ROOT)
import numpy as np, vio_ct
import test_gpu_depth_holes as T
P=vio_ct.pkg()
def run(name, cfg, sc, seq, n, frames):
    ls_o=[]; ls_h=[]
    lm_o=[]; lm_h=[]
    def hook_o(f, orc): ls_o.append(orc.line_search_stats()+orc.bound_stats()); lm_o.append(orc.landmarks_ex())
    o = vio_ct.run_oracle_sequence(cfg, sc, seq, n, frames=frames, hook=hook_o)
    b, traj, stat = vio_ct.run_hip_batch(P, cfg, sc, [seq], n, [frames], hook=lambda f, bb: (ls_h.append(bb.bound_stats(0)), lm_h.append(bb.landmarks_ex(0))))
    po=np.array([x[1] for x in o["traj"]]); ph=np.array([x[1] for x in traj[0]])
    f0=o["traj"][0][0]
    print(name, "first traj frame", f0)
    for f in range(n):
        so, sh = o["status"][f], stat[0][f]
        lo=ls_o[f]; lh=ls_h[f]
        d = np.abs(po[f-f0]-ph[f-f0]).max() if f>=f0 and f-f0<min(len(po),len(ph)) else -1
        dd = -1.0
        a, h = lm_o[f], lm_h[f]
        if a.shape == h.shape and len(a):
            have = (a[:, 3] > 0) & (h[:, 3] > 0)
            if have.any(): dd = float((np.abs(a[have, 3] - h[have, 3]) / np.abs(a[have, 3])).max())
        flag = "" if (int(so["iterations"]),int(so["successful_steps"]),int(so["n_landmarks"]))==(sh.iterations,sh.successful_steps,sh.n_landmarks) and (lo[0],lo[1],lo[3])==(lh[2],lh[3],lh[1]) else " <<<<"
        print(f, "it/succ/nlm O", int(so["iterations"]),int(so["successful_steps"]),int(so["n_landmarks"]), "H", sh.iterations,sh.successful_steps,sh.n_landmarks, "| ls(ev,con) clamps bounded O", lo, "H", (lh[2],lh[3],lh[0],lh[1]), "posdiff %.2e"%d, "worst rel depth diff %.2e"%dd, "ovf", sh.overflow_flags, flag)
if len(sys.argv) > 1 and sys.argv[1] == "near":
    cfg = P.canonical_config(); sc = vio_ct.synth_like(cfg); seq, n = 5, 60
    def edit(f, g, d):
        d[:, :160] = 100
        d[300:, 400:] = 0
        return T._moving_patch(f, g, d)
    run("near", cfg, sc, seq, n, T._frames(P, sc, seq, n, edit))
else:
    cfg, sc, seq, n, frames = T._bound_scene(P)
    run("bound", cfg, sc, seq, n, frames)
