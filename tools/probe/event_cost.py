"""What a hipEventRecord / hipStreamWaitEvent between two dependent kernels of one stream costs on this platform (measurement aid):
N ~20 us kernels back to back (the host stays ahead: differences are the device side), then with an event recorded behind every one,
then with a second stream waiting on each -- for events created with the default (system-scope) release and with
hipEventReleaseToDevice."""
import ctypes, time, torch
hip = ctypes.CDLL("libamdhip64.so")
x = torch.zeros(16 << 20, device="cuda")
N = 300
DISABLE_TIMING, REL_DEVICE, REL_SYSTEM = 0x2, 0x40000000, 0x80000000
def mk(flags):
    ev = ctypes.c_void_p()
    assert hip.hipEventCreateWithFlags(ctypes.byref(ev), ctypes.c_uint(flags)) == 0
    return ev
def run(mode, flags):
    evs = [mk(flags) for _ in range(N)]
    side = torch.cuda.Stream()
    main = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    sid = ctypes.c_void_p(side.cuda_stream)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(N):
        x.add_(1.0)
        if mode >= 1:
            hip.hipEventRecord(evs[i], main)
        if mode == 2:
            hip.hipStreamWaitEvent(sid, evs[i], 0)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / N * 1e6
    for e in evs:
        hip.hipEventDestroy(e)
    return el
run(0, DISABLE_TIMING)
for rep in range(3):
    print("us per kernel: plain %.2f | default events: +record %.2f  +record+side wait %.2f | release-to-device events: +record %.2f  +record+side wait %.2f"
          % (run(0, DISABLE_TIMING), run(1, DISABLE_TIMING), run(2, DISABLE_TIMING), run(1, DISABLE_TIMING | REL_DEVICE), run(2, DISABLE_TIMING | REL_DEVICE)))
