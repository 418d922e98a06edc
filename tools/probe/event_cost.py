"""What a hipEventRecord / hipStreamWaitEvent between two dependent kernels of one stream costs on this platform (measurement aid):
N tiny kernels back to back, then the same with an event recorded behind every one, then with a second stream waiting on each."""
import time, torch
x = torch.zeros(16 << 20, device="cuda")  # (a ~25 us kernel: the host stays ahead, the difference is the device side)
N = 400
def run(mode):
    evs = [torch.cuda.Event() for _ in range(N)]
    side = torch.cuda.Stream()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(N):
        x.add_(1.0)
        if mode >= 1:
            evs[i].record()
        if mode == 2:
            side.wait_event(evs[i])
        if mode == 3:  # main waits on an event the side stream recorded long ago
            torch.cuda.current_stream().wait_event(evs[0])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N * 1e6
for rep in range(3):
    print("us per kernel: plain %.2f  +record %.2f  +record+side wait %.2f  +record+self wait on old event %.2f" % (run(0), run(1), run(2), run(3)))
