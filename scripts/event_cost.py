"""What an event record (the fork of the weight-gradient stream) costs the stream it is recorded on: N small dependent kernels back
to back, with and without a hipEventRecord between them; default (release-to-system) events against hipEventReleaseToDevice ones,
and the full fork (record + hipStreamWaitEvent on a second stream that then runs a kernel)."""
import ctypes, time, torch
hip = ctypes.CDLL("libamdhip64.so")
hipEventDisableTiming, hipEventReleaseToDevice = 0x2, 0x40000000
def make(flags):
    e = ctypes.c_void_p(); assert hip.hipEventCreateWithFlags(ctypes.byref(e), ctypes.c_uint(flags)) == 0; return e
x = torch.zeros(1 << 16, device="cuda"); y = torch.zeros(1 << 16, device="cuda")
main = torch.cuda.current_stream(); side = torch.cuda.Stream()
sm, ss = ctypes.c_void_p(main.cuda_stream), ctypes.c_void_p(side.cuda_stream)
N = 2000
def run(mode, flags=hipEventDisableTiming):
    evs = [make(flags) for _ in range(64)]
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    big = torch.zeros(1 << 28, device="cuda")
    for _ in range(40): big.add_(1.0)              # keep the GPU busy while the host enqueues (the host must be ahead)
    a.record()
    for i in range(N):
        x.add_(1.0)
        if mode >= 1: hip.hipEventRecord(evs[i % 64], sm)
        if mode >= 2:
            hip.hipStreamWaitEvent(ss, evs[i % 64], 0)
            with torch.cuda.stream(side): y.add_(1.0)
    b.record(); torch.cuda.synchronize()
    for e in evs: hip.hipEventDestroy(e)
    return a.elapsed_time(b) * 1e3 / N
for name, mode, fl in (("kernels only", 0, 0), ("+ record (system scope)", 1, hipEventDisableTiming),
                       ("+ record (device scope)", 1, hipEventDisableTiming | hipEventReleaseToDevice),
                       ("+ fork to a side stream (system scope)", 2, hipEventDisableTiming),
                       ("+ fork to a side stream (device scope)", 2, hipEventDisableTiming | hipEventReleaseToDevice)):
    t = [run(mode, fl) for _ in range(3)]
    print(f"{name:42s} {min(t):6.2f} us per kernel on the main stream (runs: {' '.join(f'{v:.2f}' for v in t)})", flush=True)
