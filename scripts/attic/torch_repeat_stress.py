"""Control for profiles/r05_determinism.txt: do PyTorch's own kernels (MIOpen conv, rocBLAS GEMM) return the same bits when the same
call is repeated while other processes load the GPU?  A chain of conv2d + matmul on fixed inputs, enqueued in bursts, outputs compared
with the first repetition.  usage: torch_repeat_stress.py [reps] [burst]"""
import sys
import torch
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
burst = int(sys.argv[2]) if len(sys.argv) > 2 else 32
torch.manual_seed(0)
dev = 'cuda:0'
x = torch.randn(16, 16, 64, 64, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
w1 = torch.randn(16, 16, 3, 3, device=dev, dtype=torch.bfloat16) * 0.1
w2 = torch.randn(32, 16, 3, 3, device=dev, dtype=torch.bfloat16) * 0.1
a = torch.randn(1024, 1024, device=dev, dtype=torch.bfloat16)
b = torch.randn(1024, 1024, device=dev, dtype=torch.bfloat16)


def chain():
    y = torch.nn.functional.conv2d(x, w1, padding=1)
    y = torch.nn.functional.conv2d(torch.relu(y), w2, padding=1)
    return y, a @ b


ref = [t.clone() for t in chain()]
torch.cuda.synchronize()
bad = [0, 0]
for r in range(0, reps, burst):
    outs = [chain() for _ in range(burst)]
    torch.cuda.synchronize()
    for o in outs:
        for k in range(2):
            if not torch.equal(o[k].view(torch.int16), ref[k].view(torch.int16)):
                bad[k] += 1
print('%d repetitions of conv2d-relu-conv2d + matmul (bf16): conv output differs %d times, matmul output %d times' % (reps, bad[0], bad[1]))
