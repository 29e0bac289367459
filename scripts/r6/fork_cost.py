"""What one lane fork costs the main stream: a chain of 200 dependent small kernels on the main stream, alone / with an event record
between every two / with the record AND a second stream that waits for it and runs a small kernel (the step's weight-gradient fork)."""
import torch, time
dev = 'cuda:0'
x = torch.zeros(1 << 25, device=dev)           # 128 MB: ~55 us per add_, the host runs ahead
y = torch.zeros(1 << 20, device=dev)
main = torch.cuda.current_stream()
side = torch.cuda.Stream()
evs = [torch.cuda.Event() for _ in range(256)]
N = 200


def run(mode):
    for i in range(N):
        x.add_(1.0)
        if mode >= 1:
            evs[i].record(main)
        if mode >= 2:
            side.wait_event(evs[i])
            with torch.cuda.stream(side):
                y.add_(1.0)
    if mode >= 2:
        main.wait_stream(side)


for mode, name in ((0, 'chain alone'), (1, 'event record between kernels'), (2, 'record + side stream waits and runs a kernel')):
    for _ in range(3):
        run(mode)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); run(mode); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print('%-48s %.2f us per kernel' % (name, best / N * 1e6))
