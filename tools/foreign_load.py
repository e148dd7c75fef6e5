"""The three-chain step beside a foreign stream that keeps every CU busy (bf16 GEMMs of a learner, say): env-steps/s without and with the
load, the step's stream ordering and status words.  GPU only, diagnostics.   python tools/foreign_load.py [B] [steps]"""
import sys, os, time, ctypes, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_car_racing_amd.vec_env import VecMultiCarRacing
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 600
N = 2
env = VecMultiCarRacing(B, N, seed=0, auto_reset=True, streams=2)
env.reset()
g = torch.Generator(device="cuda"); g.manual_seed(1)
pool = torch.rand((64, B, N, 3), device="cuda", generator=g); pool[..., 0] = pool[..., 0] * 2 - 1
st = torch.cuda.current_stream()
def run(load):
    x = torch.randn((8192, 8192), dtype=torch.bfloat16, device="cuda")
    side = torch.cuda.Stream(); evs = []; gemms = 0
    for k in range(100): env.step(pool[k % 64])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(steps):
        if load:
            with torch.cuda.stream(side):
                while len(evs) < 6:
                    y = x @ x; e = torch.cuda.Event(); e.record(side); evs.append(e); gemms += 1
            evs = [e for e in evs if not e.query()]
        env.step(pool[k % 64])
        if k % 16 == 15: st.synchronize()
    st.synchronize(); dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    tf = gemms * 2 * 8192 ** 3 / dt / 1e12
    return B * steps / dt, tf
a, _ = run(False)
b, tf = run(True)
c, _ = run(False)
print(f"B={B} N={N}: alone {a / 1e6:.2f} M env-steps/s; beside a foreign stream of 8192^3 bf16 GEMMs {b / 1e6:.2f} M env-steps/s ({tf:.0f} TFLOP/s of GEMM went through beside it); alone again {c / 1e6:.2f} M")
print("ordering for the caller's stream:", int(env.L.mcr_step_ordering_for(env.h, ctypes.c_void_p(st.cuda_stream))), "status words:", env.status_words().tolist(), "verdict mismatches:", env.verdict_mismatches(), "frozen env-steps:", int(env.debug_counters()[3]))
env.close()
