"""Which torch streams are really concurrent on this stack?  A long sleep kernel on stream a, a tiny op + event on stream b:
if b's event completes long before a's sleep does, a and b sit on different hardware queues."""
import time, torch
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
x = torch.zeros(1024, device=dev)
streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(9)]
names = ['default'] + ['s%d' % i for i in range(1, 10)]
SLEEP = 20_000_000          # ~10 ms of spinning
def independent(a, b):
    torch.cuda.synchronize()
    with torch.cuda.stream(a):
        torch.cuda._sleep(SLEEP)
    e = torch.cuda.Event()
    with torch.cuda.stream(b):
        x.add_(1.0)
        e.record(b)
    t0 = time.perf_counter()
    e.synchronize()
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    return dt < 2e-3
for s in streams:                       # touch every stream once (queue binding may be lazy)
    with torch.cuda.stream(s):
        x.add_(1.0)
torch.cuda.synchronize()
print('rows: sleeper, cols: prober; 1 = concurrent')
print('        ' + ' '.join('%7s' % n for n in names))
for i, a in enumerate(streams):
    print('%7s ' % names[i] + ' '.join('%7s' % ('-' if i == j else int(independent(a, b))) for j, b in enumerate(streams)))
