"""Per-phase clock stamps of k_dynamics wavefronts (debug bit 8) in a steady-state rollout with the contact side
stream: where does the slowest wavefront of each launch spend its time?  GPU only, diagnostics."""
import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_car_racing_amd.vec_env import VecMultiCarRacing
from multi_car_racing_amd import _lib
B, N = 4096, int(os.environ.get("N", "2"))
env = VecMultiCarRacing(B, N, seed=0, auto_reset=True, streams=2)
env.reset()
g = torch.Generator(device="cuda"); g.manual_seed(1)
pool = torch.rand((64, B, N, 3), device="cuda", generator=g); pool[..., 0] = pool[..., 0] * 2 - 1
if int(os.environ.get("DRIVE", "0")): pool[..., 0] *= 0.1; pool[..., 1] = 1.0; pool[..., 2] = 0.0      # bench.py --actions drive
G = 1
while G < N: G *= 2
nb = (B * G + 63) // 64
ns = (B + 1) // 2                      # side stream: 2 envs per wavefront
buf = np.zeros((nb + 2 * ns) * 8, np.uint64)
_lib.check(env.L.mcr_debug_set(env.h, 256))
names = ["load+Car.step+contact init", "velocity sweeps", "position loop", "sleep+bookkeeping+epilogue"]
acc = {0: [], 1: [], 2: []}
for k in range(900):
    env.step(pool[k % 64])
    if k >= 300 and k % 10 == 0:
        _lib.check(env.L.mcr_debug_read_dynamics_stamps(env.h, _lib.ptr(buf), len(buf)))
        allst = buf.reshape(nb + 2 * ns, 8).astype(np.int64)
        for role in (0, 1, 2):
            st_r = allst[:nb] if role == 0 else (allst[nb:nb + ns] if role == 1 else allst[nb + ns:])
            d = np.diff(st_r[:, :5], axis=1)
            tot = st_r[:, 4] - st_r[:, 0]
            if role >= 1:
                ok = (tot > 0) & (st_r[:, 0] > 0)
                if not ok.any(): continue
                tot = np.where(ok, tot, -1)
            w = int(np.argmax(tot))
            acc[role].append(np.concatenate([d[w], [tot[w]]]))
        buf[:] = 0
for role, nm in ((0, "main stream (slowest wavefront per launch)"), (1, "side stream: contact envs (slowest wavefront per launch)"), (2, "third stream: resumed envs (slowest wavefront per launch)")):
    a = np.array(acc[role], float)
    if len(a) == 0: continue
    print(nm, "- shader clocks at ~2.4 GHz shown as us; n =", len(a))
    for i, n_ in enumerate(names):
        print(f"   {n_:32s} mean {a[:, i].mean() / 2400:8.1f} us   max {a[:, i].max() / 2400:8.1f} us")
    print(f"   {'total':32s} mean {a[:, 4].mean() / 2400:8.1f} us   max {a[:, 4].max() / 2400:8.1f} us")
env.close()
