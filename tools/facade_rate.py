"""Single-env gym-style facade (BASELINE config C1: N=1, the reference's own CPU-runnable case): steps/s incl. the
host round trip of every step (action upload, device step, observation download)."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multi_car_racing_amd as M
for N in (1, 2):
    env = M.make("MultiCarRacing-v0", num_agents=N, direction="CCW", use_random_direction=False, backwards_flag=False, verbose=0)
    env.seed(0); env.reset()
    rng = np.random.RandomState(0)
    for _ in range(50): env.step(rng.uniform(-1, 1, (N, 3)))
    t0 = time.perf_counter(); n = 0
    while n < 1000:
        _, _, done, _ = env.step(rng.uniform(-1, 1, (N, 3))); n += 1
        if done: env.reset()
    dt = time.perf_counter() - t0
    print(f"facade N={N}: {n / dt:.0f} env-steps/s ({dt / n * 1e3:.3f} ms per step incl. host round trip)")
    env.close()
