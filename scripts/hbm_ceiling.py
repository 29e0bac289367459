"""What this part's HBM sustains with plain streaming kernels (torch elementwise copies / reductions, 1 GiB bf16 tensors, and the
164 MB working set of a 400x400x32 layer): the practical ceiling the HBM-bound conv kernels are compared with besides the 8 TB/s spec."""
import torch
dev = 'cuda:0'
def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for mb in (82, 164, 1024, 4096):
    n = mb * 1024 * 1024 // 2
    a = torch.empty(n, dtype=torch.bfloat16, device=dev).normal_()
    b = torch.empty_like(a)
    t = timeit(lambda: b.copy_(a))
    print('%5d MB  copy (read + write)  %7.1f us  %5.2f TB/s' % (mb, t * 1e6, 2 * n * 2 / t / 1e12))
    t = timeit(lambda: a.view(torch.int16).max())
    print('%5d MB  read-only reduction  %7.1f us  %5.2f TB/s' % (mb, t * 1e6, n * 2 / t / 1e12))
    t = timeit(lambda: b.zero_())
    print('%5d MB  write-only fill      %7.1f us  %5.2f TB/s' % (mb, t * 1e6, n * 2 / t / 1e12))
    t = timeit(lambda: torch.add(a, a, out=b))
    print('%5d MB  b = a + a            %7.1f us  %5.2f TB/s' % (mb, t * 1e6, 2 * n * 2 / t / 1e12))
    del a, b
