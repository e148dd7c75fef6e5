"""The three-chain step beside a foreign stream that keeps every CU busy (bf16 GEMMs of a learner, say): env-steps/s without and with the
load; with the load on a CU-masked stream (hipExtStreamCreateWithCUMask: k CUs of every XCD left to the env's chains) and on a LOW-priority
stream, with the env's caller stream at high priority; the shader clock in every phase, the step's stream ordering and status words.
GPU only, diagnostics.   python tools/foreign_load.py [B] [steps]"""
import sys, os, time, ctypes, subprocess, re, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_car_racing_amd.vec_env import VecMultiCarRacing

_hip = ctypes.CDLL("libamdhip64.so"); _keep = []


def masked_stream(leave_cus_per_xcd, num_xcd=8, cus_per_xcd=32):
    """a torch stream whose kernels stay off the last `leave_cus_per_xcd` CUs of every XCD (mask bit i = CU i // 8 of XCD i % 8, ubench/cumask_probe.hip)"""
    words = [0] * ((num_xcd * cus_per_xcd + 31) // 32)
    for c in range(cus_per_xcd - leave_cus_per_xcd):
        for x_ in range(num_xcd):
            i = c * num_xcd + x_; words[i // 32] |= 1 << (i % 32)
    arr = (ctypes.c_uint32 * len(words))(*words); st_ = ctypes.c_void_p()
    assert _hip.hipExtStreamCreateWithCUMask(ctypes.byref(st_), ctypes.c_uint32(len(words)), arr) == 0
    s_ = torch.cuda.ExternalStream(st_.value); _keep.append((st_, s_)); return s_


def priority_stream(prio):
    """hipStreamCreateWithPriority: -1 high, 0 normal, 1 low (where the device has three levels)"""
    st_ = ctypes.c_void_p()
    assert _hip.hipStreamCreateWithPriority(ctypes.byref(st_), ctypes.c_uint(0), ctypes.c_int(prio)) == 0
    s_ = torch.cuda.ExternalStream(st_.value); _keep.append((st_, s_)); return s_
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 600
N = 2
env = VecMultiCarRacing(B, N, seed=0, auto_reset=True, streams=2)
env.reset()
g = torch.Generator(device="cuda"); g.manual_seed(1)
pool = torch.rand((64, B, N, 3), device="cuda", generator=g); pool[..., 0] = pool[..., 0] * 2 - 1
st = torch.cuda.current_stream()
x = torch.randn((8192, 8192), dtype=torch.bfloat16, device="cuda")


def sclk():
    """current shader / memory clock as rocm-smi prints them (MHz)"""
    try:
        out = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=20).stdout
        s = re.search(r"sclk clock level:.*?\((\d+)Mhz\)", out); m = re.search(r"mclk clock level:.*?\((\d+)Mhz\)", out)
        return f"sclk {s.group(1) if s else '?'} mclk {m.group(1) if m else '?'} MHz"
    except Exception as e:
        return f"(rocm-smi: {e})"


def run(load, side=None, label="", env_stream=None):
    side = side or torch.cuda.Stream()
    es = env_stream or st
    evs = []; gemms = 0
    with torch.cuda.stream(es):
        for k in range(100): env.step(pool[k % 64])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    clk = None
    for k in range(steps):
        if load:
            with torch.cuda.stream(side):
                while len(evs) < 6:
                    y = x @ x; e = torch.cuda.Event(); e.record(side); evs.append(e); gemms += 1
            evs = [e for e in evs if not e.query()]
        with torch.cuda.stream(es):
            env.step(pool[k % 64])
        if k % 64 == 63: es.synchronize()
        if k == steps // 2: clk = sclk()
    es.synchronize(); dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    tf = gemms * 2 * 8192 ** 3 / dt / 1e12
    print(f"  {label:58s} {B * steps / dt / 1e6:6.2f} M env-steps/s   GEMM {tf:6.0f} TFLOP/s   [{clk}]   ordering {int(env.L.mcr_step_ordering_for(env.h, ctypes.c_void_p(es.cuda_stream)))}", flush=True)
    return B * steps / dt, tf


def gemm_only(side, label):
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
    with torch.cuda.stream(side):
        for _ in range(60): y = x @ x; n += 1
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"  {label:46s} GEMM alone {n * 2 * 8192 ** 3 / dt / 1e12:6.0f} TFLOP/s   [{sclk()}]", flush=True)


lo, hi = ctypes.c_int(), ctypes.c_int()
_hip.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi))
print(f"# tools/foreign_load.py: B={B} N={N}, {steps} steps per leg, the caller fences every 64 steps; stream priorities least {lo.value} .. greatest {hi.value}")
run(False, label="alone"); run(False, label="alone (again: run-to-run spread)"); run(False, label="alone (third)")
gemm_only(torch.cuda.Stream(), "plain stream (first call: library start-up)"); gemm_only(torch.cuda.Stream(), "plain stream")
run(True, label="beside 8192^3 bf16 GEMMs, plain stream")
run(False, label="alone, right after the load")
time.sleep(8)
run(False, label="alone, 8 s later")
for k in (4,):
    s = masked_stream(k)
    gemm_only(s, f"CU-masked stream (leaves {k} CUs per XCD)")
    run(True, side=s, label=f"beside the GEMMs on a CU-masked stream ({k} per XCD left)")
low = priority_stream(lo.value)
gemm_only(low, f"low-priority stream ({lo.value})")
run(True, side=low, label="beside the GEMMs on a LOW-priority stream")
high = priority_stream(hi.value)
run(False, env_stream=high, label="alone, env stepped on a HIGH-priority caller stream")
run(True, env_stream=high, label="beside plain-stream GEMMs, env on a HIGH-priority stream")
run(True, side=low, env_stream=high, label="GEMMs low priority + env on a high-priority stream")
run(False, label="alone, at the end")
print("status words:", env.status_words().tolist(), "verdict mismatches:", env.verdict_mismatches(), "frozen env-steps:", int(env.debug_counters()[3]))
env.close()
